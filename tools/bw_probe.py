#!/usr/bin/env python3
"""Calibration: achieved HBM GB/s of the bandwidth-bound kernels against a plain device copy."""
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


for (n, h, w, c, s) in [(256, 112, 112, 32, 1), (256, 112, 112, 96, 2), (256, 56, 56, 144, 1), (256, 28, 28, 192, 1), (256, 14, 14, 576, 1)]:
    x = torch.randn((n, h, w, c), device=dev)
    wt = torch.randn((3, 3, c), device=dev)
    sc, bi = torch.rand(c, device=dev), torch.rand(c, device=dev)
    ms = t(lambda: ops.dwconv3x3_bn_act(x, wt, sc, bi, s))
    oh = (h - 1) // s + 1
    byt = 4.0 * n * c * (h * w + oh * oh)
    y = torch.empty_like(x)
    msc = t(lambda: y.copy_(x))
    print("dw %dx%dx%d s%d: %.3f ms  %.0f GB/s   | torch copy of the input: %.3f ms %.0f GB/s" %
          (h, w, c, s, ms, byt / ms / 1e6, msc, 8.0 * x.numel() / msc / 1e6), flush=True)
x = torch.randn((1024, 48, 48, 64), device=dev)
ms = t(lambda: ops.maxpool3x3s2(x))
print("maxpool 48x48x64 x1024: %.3f ms %.0f GB/s" % (ms, 4.0 * x.numel() * 1.25 / ms / 1e6))
