#!/usr/bin/env python3
"""Small-batch conv form (tile id 95, csrc/conv_lat.hip: v_mfma_f32_16x16x4_f32, 32x32 tiles) against the engine's automatic
choice on every conv shape of the ResNet-50 trunk at n patches of 96^2 (default 16 = BASELINE config 1's B*T): equality + time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import hip_ops as ops  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16


def timeit(fn, iters=20):
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


cases = []
hw, inpl = 24, 64
for s, (planes, nb) in enumerate(((64, 3), (128, 4), (256, 6), (512, 3))):
    for b in range(nb):
        stride = 2 if (b == 0 and s > 0) else 1
        cases.append(("L%d.%d.c1" % (s + 1, b), hw, inpl, planes, 1, 1, 0, False, ops.ACT_RELU))
        cases.append(("L%d.%d.c2" % (s + 1, b), hw, planes, planes, 3, stride, 1, False, ops.ACT_RELU))
        oh = (hw + 2 - 3) // stride + 1
        if b == 0:
            cases.append(("L%d.%d.ds" % (s + 1, b), hw, inpl, planes * 4, 1, stride, 0, False, ops.ACT_NONE))
        cases.append(("L%d.%d.c3" % (s + 1, b), oh, planes, planes * 4, 1, 1, 0, True, ops.ACT_RELU))
        hw, inpl = oh, planes * 4
seen, ta_sum, tb_sum, all_eq = set(), 0.0, 0.0, True
for name, hw, cin, cout, k, stride, pad, res, act in cases:
    key = (hw, cin, cout, k, stride, res)
    x = torch.randn((n, hw, hw, cin), device=dev)
    w = torch.randn((cout, k, k, cin), device=dev) * 0.05
    sc, bi = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    oh = (hw + 2 * pad - k) // stride + 1
    r = torch.randn((n, oh, oh, cout), device=dev) if res else None
    a = ops.conv2d_bn_act(x, w, sc, bi, r, stride, pad, act)
    b = ops.conv2d_bn_act(x, w, sc, bi, r, stride, pad, act, tile=95)
    eq = torch.equal(a, b)
    all_eq &= eq
    ta = timeit(lambda: ops.conv2d_bn_act(x, w, sc, bi, r, stride, pad, act))
    tb = timeit(lambda: ops.conv2d_bn_act(x, w, sc, bi, r, stride, pad, act, tile=95))
    ta_sum += ta
    tb_sum += tb
    if key not in seen:
        seen.add(key)
        print("%-10s %2d^2 %4d -> %4d k%d s%d: M=%5d K=%4d N=%4d  engine %6.1f us | latency form %6.1f us | equal %s" % (
            name, hw, cin, cout, k, stride, n * oh * oh, k * k * cin, cout, ta * 1e3, tb * 1e3, eq))
print("sum over the %d conv launches of the trunk (stem excluded), %d patches: engine %.3f ms, latency form %.3f ms; all equal: %s" % (
    len(cases), n, ta_sum, tb_sum, all_eq))
