#!/usr/bin/env python3
"""Small-batch step time under different trunk plans: fused stage-1 launches on / off x row threshold of the small-batch conv form."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_extras import act_args, synth_model_state
from adafocus_amd import synth
from adafocus_amd.gfv_net import GFV
dev = torch.device("cuda:0")
for b, t in ((1, 8), (2, 8), (2, 16), (4, 16)):
    m = GFV(act_args(t, 96, b)).eval()
    m.load_state_dict(synth_model_state(m, 1007), strict=True)
    m = m.to(dev)
    fr = torch.from_numpy(synth.synth_frames(b, t, 224, seed=1)).to(dev).view(b * t, 3, 224, 224)
    act = torch.from_numpy(synth.synth_actions(b * t, 7, seed=2)[1]).to(dev)
    gv = torch.randn((b, t, 1280), device=dev)
    trunk = m.focuser.net._sync()
    out, ref = [], None
    with torch.no_grad():
        for fusion in (1, 0):
            for rows in (1536,):
                trunk.set_fusion(fusion)
                trunk.set_latency_rows(rows)
                for _ in range(5):
                    lg = m.hot_path(fr, gv, act, b, t)[0]
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(200):
                    lg = m.hot_path(fr, gv, act, b, t)[0]
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) * 5
                if ref is None:
                    ref = lg.clone()
                out.append("f%d/%d: %.3f%s" % (fusion, rows, ms, "" if torch.equal(lg, ref) else " (BITS DIFFER)"))
    print("B%d T%d (%d patches): " % (b, t, b * t) + "  ".join(out), flush=True)
    del m
