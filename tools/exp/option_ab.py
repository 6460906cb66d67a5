#!/usr/bin/env python3
"""Same-box A/B of a global option of the library on the ResNet-50 trunk: alternating runs, ms per pass.
usage: option_ab.py <option> [P=96] [n=1024] [segments=0] [math=f32]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adafocus_amd import _lib, synth  # noqa: E402
from adafocus_amd.resnet import resnet50  # noqa: E402

dev = torch.device("cuda:0")
opt = sys.argv[1]
p = int(sys.argv[2]) if len(sys.argv) > 2 else 96
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
seg = int(sys.argv[4]) if len(sys.argv) > 4 else 0
math = sys.argv[5] if len(sys.argv) > 5 else "f32"
net = resnet50(num_classes=200).eval()
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1).items()})
net = net.to(dev)
net.set_math(math)
x = torch.randn((n, p, p, 4), device=dev)
x[..., 3] = 0
trunk = net._sync()
fw = (lambda: trunk.forward(x, tsm_segments=seg)) if seg else (lambda: trunk.forward(x))


def ms(reps=20):
    for _ in range(3):
        fw()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fw()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


outs = {}
for v in (1, 0):
    with _lib.option(opt, v):
        outs[v] = fw().clone()
print("bit-identical:", torch.equal(outs[0], outs[1]))
res = {0: [], 1: []}
for rnd in range(4):
    for v in (1, 0):
        with _lib.option(opt, v):
            res[v].append(ms())
for v in (1, 0):
    print("%s=%d  P=%d n=%d T=%d %s trunk ms: %s  min %.3f" % (opt, v, p, n, seg, math, " ".join("%.3f" % t for t in res[v]), min(res[v])))
