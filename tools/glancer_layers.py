#!/usr/bin/env python3
"""Per-layer micro-benchmark of the glancer's MobileNetV2 at one 256-frame chunk: each expand / depthwise / project
launch timed on its own with HIP events, next to its algorithmic HBM bytes (in + out + residual) and FLOPs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import hip_ops as ops  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]


def timeit(fn, iters=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def conv(name, hw, cin, cout, k, stride, act, res=False, cin_pad=None):
    cp = cin_pad or cin
    x = torch.randn((n, hw, hw, cp), device=dev)
    w = torch.randn((cout, k, k, cp), device=dev) * 0.05
    sc, bi = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    oh = (hw + 2 * (k // 2) - k) // stride + 1
    r = torch.randn((n, oh, oh, cout), device=dev) if res else None
    out = torch.empty((n, oh, oh, cout), device=dev)
    ms = timeit(lambda: ops.conv2d_bn_act(x, w, sc, bi, r, stride, k // 2, act, out=out))
    by = 4.0 * n * (hw * hw * cin + oh * oh * cout * (2 if res else 1))
    fl = 2.0 * n * oh * oh * cout * k * k * cin
    return name, ms, by, fl


def dw(name, hw, c, stride):
    x = torch.randn((n, hw, hw, c), device=dev)
    w = torch.randn((3, 3, c), device=dev)
    sc, bi = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    ms = timeit(lambda: ops.dwconv3x3_bn_act(x, w, sc, bi, stride))
    oh = (hw + 2 - 3) // stride + 1
    return name, ms, 4.0 * n * c * (hw * hw + oh * oh), 2.0 * n * oh * oh * c * 9


rows = [conv("stem 3x3s2 3->32", 224, 3, 32, 3, 2, ops.ACT_RELU6, cin_pad=4)]
hw, cin, bi_ = 112, 32, 1
for t, c, reps, s in cfg:
    for i in range(reps):
        stride = s if i == 0 else 1
        hid = cin * t
        tag = "b%d" % bi_
        if t != 1:
            rows.append(conv(tag + " expand %d->%d @%d" % (cin, hid, hw), hw, cin, hid, 1, 1, ops.ACT_RELU6))
        rows.append(dw(tag + " dw %d @%d s%d" % (hid, hw, stride), hw, hid, stride))
        ohw = (hw + 2 - 3) // stride + 1
        rows.append(conv(tag + " project %d->%d @%d" % (hid, c, ohw), ohw, hid, c, 1, 1, ops.ACT_NONE, res=(stride == 1 and cin == c)))
        hw, cin, bi_ = ohw, c, bi_ + 1
rows.append(conv("head 320->1280 @7", 7, 320, 1280, 1, 1, ops.ACT_RELU6))
tot = roof = 0.0
print("%-34s %8s %8s %8s %9s" % ("launch", "ms", "TB/s", "TF", "ms@roof"))
for name, ms, by, fl in rows:
    r = max(by / 5.5e12, fl / 150e12) * 1e3
    tot += ms
    roof += r
    print("%-34s %8.4f %8.2f %8.1f %9.4f" % (name, ms, by / ms / 1e9, fl / ms / 1e9, r))
print("total %.3f ms for %d frames (x%.0f for 1024: %.2f ms); roof-sum %.3f ms" % (tot, n, 1024 / n, tot * 1024 / n, roof))
