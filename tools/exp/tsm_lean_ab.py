#!/usr/bin/env python3
"""Same-box A/B of the lean K loop for the temporally shifted conv1 launches (option "tsm_lean"): ResNet-50 trunk with T segments, alternating runs.
usage: tsm_lean_ab.py [P=128] [n=512] [segments=8]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adafocus_amd import _lib, synth  # noqa: E402
from adafocus_amd.resnet import resnet50  # noqa: E402

dev = torch.device("cuda:0")
p = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
seg = int(sys.argv[3]) if len(sys.argv) > 3 else 8
net = resnet50(num_classes=200).eval()
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1).items()})
net = net.to(dev)
x = torch.randn((n, p, p, 4), device=dev)
x[..., 3] = 0
trunk = net._sync()


def ms(reps=20):
    for _ in range(3):
        trunk.forward(x, tsm_segments=seg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        trunk.forward(x, tsm_segments=seg)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


res = {0: [], 1: []}
for rnd in range(4):
    for lean in (1, 0):
        with _lib.option("tsm_lean", lean):
            res[lean].append(ms())
for lean in (1, 0):
    print("tsm_lean=%d  P=%d n=%d T=%d trunk ms: %s  min %.3f" % (lean, p, n, seg, " ".join("%.3f" % v for v in res[lean]), min(res[lean])))
