#!/usr/bin/env python3
"""Glancer tail 1x1 convs (512 frames): every engine tile id against the automatic choice."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import hip_ops as ops
dev = torch.device("cuda:0")
n = 512
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
cases = [("b8 expand", 14, 64, 384, False, ops.ACT_RELU6), ("b8 project", 14, 384, 64, True, ops.ACT_NONE), ("b12 expand", 14, 96, 576, False, ops.ACT_RELU6),
         ("b12 project", 14, 576, 96, True, ops.ACT_NONE), ("b15 expand", 7, 160, 960, False, ops.ACT_RELU6), ("b15 project", 7, 960, 160, True, ops.ACT_NONE),
         ("b17 project", 7, 960, 320, False, ops.ACT_NONE), ("head", 7, 320, 1280, False, ops.ACT_RELU6)]
tiles = [0, 31, 32, 33, 34, 38, 39, 71, 72, 73, 74, 21, 22, 23, 25, 26]      # (27 / 37: tools/exp/build_exp_tiles.sh)
for name, hw, k, cout, res, act in cases:
    x = torch.randn((n, hw, hw, k), device=dev); w = torch.randn((cout, 1, 1, k), device=dev) * 0.05
    sc, bi = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    r = torch.randn((n, hw, hw, cout), device=dev) if res else None
    out = torch.empty((n, hw, hw, cout), device=dev)
    fl = 2.0 * n * hw * hw * k * cout
    row = []
    for t in tiles:
        try:
            ms = timeit(lambda: ops.conv2d_bn_act(x, w, sc, bi, r, 1, 0, act, tile=t, out=out))
            row.append((ms, t))
        except Exception:
            pass
    auto = [m for m, t in row if t == 0][0]
    best = min(row)
    print("%-12s K=%3d N=%4d: auto %.1f us (%.0f TF) | best tile %d: %.1f us (%.0f TF) | " % (name, k, cout, auto * 1e3, fl / auto / 1e9, best[1], best[0] * 1e3, fl / best[0] / 1e9)
          + " ".join("%d:%.0f" % (t, m * 1e3) for m, t in row))
