#!/usr/bin/env python3
"""Per-launch table of the ResNet-50 trunk (HIP-event bracketed): ms, TFLOP/s, algorithmic TB/s, tile.
usage: layer_table.py [patch=96] [patches=1024] [tsm_segments=0]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import synth  # noqa: E402
from adafocus_amd.resnet import resnet50  # noqa: E402

dev = torch.device("cuda:0")
p = int(sys.argv[1]) if len(sys.argv) > 1 else 96
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
tsm = int(sys.argv[3]) if len(sys.argv) > 3 else 0
net = resnet50(num_classes=200).eval()
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1).items()})
net = net.to(dev)
x = torch.randn((n, p, p, 4), device=dev)
x[..., 3] = 0
trunk = net._sync()
for _ in range(3):
    trunk.forward(x, tsm_segments=tsm) if tsm else trunk.forward(x)
runs = [trunk.profile(x, tsm_segments=tsm) for _ in range(5)]
names = ["stem", "maxpool"]
for li, nb in enumerate((3, 4, 6, 3), 1):
    for b in range(nb):
        names += ["L%d.%d.c1" % (li, b), "L%d.%d.c2" % (li, b)]
        if b == 0:
            names.append("L%d.%d.ds" % (li, b))
        names.append("L%d.%d.c3" % (li, b))
names.append("avgpool")
tot = 0.0
ideal = 0.0
print("%-10s %8s %8s %8s %5s %9s" % ("launch", "ms", "TF", "TB/s", "tile", "ms@roof"))
for i, nm in enumerate(names):
    ms = min(r[i]["ms"] for r in runs)
    e = runs[0][i]
    roof = max(e["flops"] / 150e12, e["bytes"] / 5.5e12) * 1e3
    tot += ms
    ideal += roof
    print("%-10s %8.4f %8.1f %8.2f %5d %9.4f" % (nm, ms, e["flops"] / ms / 1e9, e["bytes"] / ms / 1e9, e["tile"], roof))
print("total %.3f ms (sum of per-launch minima), roof-sum %.3f ms" % (tot, ideal))
