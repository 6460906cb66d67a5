#!/usr/bin/env python3
"""Small-batch latency of the hot path (gather -> ResNet-50 -> GRU classifier): launches issued one by one from Python
(hot_path) vs the same step replayed from a captured HIP graph (GFV.capture_hot_path).  B = 1, 2 at T = 8 (BASELINE config
1 is B = 2, T = 8, P = 96) and T = 16."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import act_args, synth_model_state  # noqa: E402
from adafocus_amd import synth  # noqa: E402
from adafocus_amd.gfv_net import GFV  # noqa: E402


def latency_rows(dev, cases=((1, 8), (2, 8), (2, 16)), p=96, iters=200):
    rows = {}
    for b, t in cases:
        m = GFV(act_args(t, p, b)).eval()
        m.load_state_dict(synth_model_state(m, 1007), strict=True)
        m = m.to(dev)
        fr = torch.from_numpy(synth.synth_frames(b, t, 224, seed=1)).to(dev).view(b * t, 3, 224, 224)
        act = torch.from_numpy(synth.synth_actions(b * t, 7, seed=2)[1]).to(dev)
        gv = torch.randn((b, t, 1280), device=dev)
        with torch.no_grad():
            for _ in range(5):
                lg = m.hot_path(fr, gv, act, b, t)[0]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                lg = m.hot_path(fr, gv, act, b, t)[0]
            torch.cuda.synchronize()
            eager = (time.perf_counter() - t0) / iters
            ref = lg.clone()
            g = m.capture_hot_path(b, t)
            g(fr, gv, act)
            torch.cuda.synchronize()
            same = bool(torch.equal(g.logits, ref))
            t0 = time.perf_counter()
            for _ in range(iters):
                g.replay()
            torch.cuda.synchronize()
            graph = (time.perf_counter() - t0) / iters
        rows["B%d_T%d_P%d" % (b, t, p)] = {"eager_ms": round(eager * 1e3, 4), "graph_ms": round(graph * 1e3, 4),
                                            "speedup": round(eager / graph, 2), "clips_per_s_graph": round(b / graph, 1),
                                            "bit_identical": same}
        del g, m
    return rows


if __name__ == "__main__":
    print(json.dumps(latency_rows(torch.device("cuda:0"))))
