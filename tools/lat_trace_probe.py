#!/usr/bin/env python3
"""One small-batch hot-path step (B = 1, T = 8, P = 96) repeated: for a rocprofv3 --kernel-trace run; tools/lat_trace_report.py turns the
trace into a per-launch timeline of one step (duration of each kernel and the gap in front of it)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import synth  # noqa: E402
from adafocus_amd.gfv_net import GFV  # noqa: E402
from bench_extras import act_args, synth_model_state  # noqa: E402

dev = torch.device("cuda:0")
b, t = int(os.environ.get("B", "1")), 8
m = GFV(act_args(t, 96, b)).eval()
m.load_state_dict(synth_model_state(m, 1007), strict=True)
m = m.to(dev)
fr = torch.from_numpy(synth.synth_frames(b, t, 224, seed=1)).to(dev).view(b * t, 3, 224, 224)
act = torch.from_numpy(synth.synth_actions(b * t, 7, seed=2)[1]).to(dev)
gv = torch.randn((b, t, 1280), device=dev)
with torch.no_grad():
    for _ in range(60):
        m.hot_path(fr, gv, act, b, t)
torch.cuda.synchronize()
