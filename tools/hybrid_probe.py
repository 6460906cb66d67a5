#!/usr/bin/env python3
"""Premise check for hybrid-tile launches: big 128x128 tiles on as many full CU-rounds of rows as fit,
small 64x64 tiles on the remainder, vs. one uniform launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import hip_ops as ops  # noqa: E402

dev = torch.device("cuda:0")


def t(fn, it=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


cases = [("L3.1.c2", 1024, 6, 6, 256, 256, 3, 1), ("L3.1.c1", 1024, 6, 6, 1024, 256, 1, 0), ("L4.1.c2", 1024, 3, 3, 512, 512, 3, 1),
         ("L4.1.c3", 1024, 3, 3, 512, 2048, 1, 0), ("L2.1.c2", 1024, 12, 12, 128, 128, 3, 1), ("L2.1.c1", 1024, 12, 12, 512, 128, 1, 0)]
for name, n, h, w, cin, cout, k, pad in cases:
    x = torch.randn((n, h, w, cin), device=dev)
    wt = torch.randn((cout, k, k, cin), device=dev) * 0.05
    sc = torch.rand(cout, device=dev) + 0.5
    bi = torch.randn(cout, device=dev) * 0.1
    flops = 2.0 * n * h * w * cout * k * k * cin
    base = {tl: t(lambda: ops.conv2d_bn_act(x, wt, sc, bi, None, 1, pad, 1, tile=tl)) for tl in (21, 22, 23)}
    M = n * h * w
    tiles_n = (cout + 127) // 128
    rounds = (M // 128) * tiles_n // 256
    rows_big = rounds * 256 // tiles_n * 128
    n_big = rows_big // (h * w)
    if n_big == 0 or n_big == n:
        print(name, "no split", base)
        continue
    xb, xs = x[:n_big].contiguous(), x[n_big:].contiguous()
    tb = t(lambda: ops.conv2d_bn_act(xb, wt, sc, bi, None, 1, pad, 1, tile=21))
    ts = {tl: t(lambda: ops.conv2d_bn_act(xs, wt, sc, bi, None, 1, pad, 1, tile=tl)) for tl in (22, 23)}
    both = t(lambda: (ops.conv2d_bn_act(xb, wt, sc, bi, None, 1, pad, 1, tile=21), ops.conv2d_bn_act(xs, wt, sc, bi, None, 1, pad, 1, tile=23)))
    print("%s uniform: %s | big(%d imgs) %.3f + small(%d imgs) %s ; sequential both %.3f ms -> %.1f TF (best uniform %.1f TF)"
          % (name, {k_: round(v, 3) for k_, v in base.items()}, n_big, tb, n - n_big, {k_: round(v, 3) for k_, v in ts.items()}, both,
             flops / both / 1e9, flops / min(base.values()) / 1e9), flush=True)
