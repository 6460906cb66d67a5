#!/usr/bin/env python3
"""Glancer tail depthwise launches (MobileNetV2 b8-b17: 14^2 x 384/576, 7^2 x 960): `dwconv3x3_kernel` against the LDS-staged,
several-frames-per-block `dw_same_kernel` written for EfficientNet (same taps, same order): equality + time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import hip_ops as ops  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for hw, c in ((28, 192), (14, 384), (14, 576), (7, 960), (56, 144)):
    x = torch.randn((n, hw, hw, c), device=dev)
    w = torch.randn((3, 3, c), device=dev)
    sc, bi = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    a = ops.dwconv3x3_bn_act(x, w, sc, bi, 1)
    b = ops.dwconv_same_bn_act(x, w, sc, bi, 3, 1, ops.ACT_RELU6)
    ta = timeit(lambda: ops.dwconv3x3_bn_act(x, w, sc, bi, 1))
    tb = timeit(lambda: ops.dwconv_same_bn_act(x, w, sc, bi, 3, 1, ops.ACT_RELU6))
    by = 8.0 * n * hw * hw * c
    print("%3d^2 x %4d, %d frames: dwconv3x3 %.1f us (%.2f TB/s) | dw_same %.1f us (%.2f TB/s) | equal %s  maxdiff %.2e" % (
        hw, c, n, ta * 1e3, by / ta / 1e9, tb * 1e3, by / tb / 1e9, torch.equal(a, b), (a - b).abs().max().item()))

print("fp16 storage (EfficientNet-B3 3x3 / stride-1 shapes at 1024 patches):")
n = 1024
for hw, c in ((72, 40), (36, 192), (9, 576), (5, 1392)):
    x = torch.randn((n, hw, hw, c), device=dev).half()
    w = torch.randn((3, 3, c), device=dev)
    sc, bi = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    a = ops.dwconv3x3_bn_act_f16(x, w, sc, bi, 1)
    b = ops.dwconv_same_bn_act(x, w, sc, bi, 3, 1, ops.ACT_RELU6)
    ta = timeit(lambda: ops.dwconv3x3_bn_act_f16(x, w, sc, bi, 1))
    tb = timeit(lambda: ops.dwconv_same_bn_act(x, w, sc, bi, 3, 1, ops.ACT_RELU6))
    tc = timeit(lambda: ops.dwconv_same_bn_act(x, w, sc, bi, 3, 1, ops.ACT_SWISH, want_pool=True))
    by = 4.0 * n * hw * hw * c
    print("%3d^2 x %4d: dwconv3x3 (8-byte loads) %.1f us (%.2f TB/s) | dw_same %.1f us (%.2f TB/s) | dw_same swish + squeeze sums %.1f us | equal %s" % (
        hw, c, ta * 1e3, by / ta / 1e9, tb * 1e3, by / tb / 1e9, tc * 1e3, torch.equal(a, b)))
