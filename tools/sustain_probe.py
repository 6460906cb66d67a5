#!/usr/bin/env python3
"""Does the conv engine's rate hold under sustained load?  Times groups of 10 launches of the 8192^2 x 4096 GEMM
(fp32 pipe tile 31, split tile 41) for ~2 s each and samples rocm-smi clocks/power in between."""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import hip_ops as ops  # noqa: E402

dev = torch.device("cuda:0")
x = torch.randn((8192, 1, 1, 4096), device=dev)
w = torch.randn((8192, 1, 1, 4096), device=dev) * 0.02
sc, bi = torch.ones(8192, device=dev), torch.zeros(8192, device=dev)
out = torch.empty((8192, 1, 1, 8192), device=dev)
fl = 2.0 * 8192 * 8192 * 4096


def smi():
    try:
        o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in o.splitlines() if ("sclk" in l or "Power" in l or "mclk" in l)]
        return " | ".join(keep)[:300]
    except Exception as e:  # noqa: BLE001
        return repr(e)


for tile in (31, 41):
    for _ in range(3):
        ops.conv2d_bn_act(x, w, sc, bi, None, 1, 0, ops.ACT_NONE, tile=tile, out=out)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
    evs[0].record()
    for g in range(60):
        for _ in range(10):
            ops.conv2d_bn_act(x, w, sc, bi, None, 1, 0, ops.ACT_NONE, tile=tile, out=out)
        evs[g + 1].record()
    print("during:", smi(), flush=True)
    torch.cuda.synchronize()
    tf = [fl * 10 / (evs[g].elapsed_time(evs[g + 1]) * 1e-3) / 1e12 for g in range(60)]
    print("tile %d TF per group of 10: first %s ... last %s" % (tile, ["%.1f" % v for v in tf[:5]], ["%.1f" % v for v in tf[-5:]]), flush=True)
print("idle:", smi())
