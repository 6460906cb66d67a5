#!/usr/bin/env python3
"""Why is a shifted conv1 on the lean K loop FASTER than the plain lean conv1?  Run the plain arithmetic through the shifted kernel: with
tsm_div > cin the fold is 0 channels, so no slice is shifted and the result equals the plain trunk's -- per-launch times of the conv1 launches of
stages 2-4 (profile()), plain kernel vs LTSM kernel, tsm_lean 1 / 0.   usage: ltsm_plain_probe.py [P=128] [n=512]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adafocus_amd import _lib, synth  # noqa: E402
from adafocus_amd.resnet import resnet50  # noqa: E402

dev = torch.device("cuda:0")
p = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
net = resnet50(num_classes=200).eval()
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1).items()})
net = net.to(dev)
x = torch.randn((n, p, p, 4), device=dev)
x[..., 3] = 0
trunk = net._sync()
plain = trunk.forward(x).clone()
same = trunk.forward(x, tsm_segments=8, tsm_div=1 << 20).clone()
print("fold 0 == plain:", torch.equal(plain, same))


def table(**kw):
    for _ in range(2):
        trunk.forward(x, **kw)
    runs = [trunk.profile(x, **kw) for _ in range(5)]
    best = [min(r[i]["ms"] for r in runs) for i in range(len(runs[0]))]
    return [(e["tile"], e["flops"], m) for e, m in zip(runs[0], best)]


a = table()
b = table(tsm_segments=8, tsm_div=1 << 20)
with _lib.option("tsm_lean", 0):
    c = table(tsm_segments=8, tsm_div=1 << 20)
print("plain plan: %d launches %.3f ms; fold-0 shifted plan (lean): %d launches %.3f ms; (select form): %.3f ms" %
      (len(a), sum(m for _, _, m in a), len(b), sum(m for _, _, m in b), sum(m for _, _, m in c)))
# stages 2-4 line up from the end (same launches in both plans)
k = 36
print("last %d launches (stage 2 tail .. stage 4): plain kernel / LTSM kernel / select kernel, ms" % k)
for (t0, f0, m0), (t1, f1, m1), (t2, f2, m2) in zip(a[-k:], b[-k:], c[-k:]):
    flag = "  <-- conv1" if abs(m0 - m1) > 0.004 else ""
    print("tile %3d %7.2f GF   %.4f  %.4f  %.4f%s" % (t0, f0 / 1e9, m0, m1, m2, flag))
