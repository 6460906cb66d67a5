#!/usr/bin/env python3
"""Split tiles with weights pre-split at load time (6x) vs split on the fly (4x): bit-equality and per-launch time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import synth  # noqa: E402
from adafocus_amd.resnet import resnet50  # noqa: E402

dev = torch.device("cuda:0")
n, p = 1024, 96
net = resnet50(num_classes=200).eval()
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1).items()})
net = net.to(dev)
net.set_math("split_bf16")
x = torch.randn((n, p, p, 4), device=dev)
x[..., 3] = 0
trunk = net._sync()
nconv = 53


def run(tiles):
    trunk.set_tiles(tiles)
    for _ in range(2):
        out = trunk.forward(x).clone()
    runs = [trunk.profile(x) for _ in range(3)]
    return out, [min(r[i]["ms"] for r in runs) for i in range(len(runs[0]))]


ref, base = run([0] * nconv)
print("auto (split math): %.3f ms" % sum(base), flush=True)
res = {}
for t in (41, 42, 61, 62, 63, 64, 65, 66, 67):
    out, res[t] = run([0] + [t] * (nconv - 1))
    print("all tile %d: %.3f ms  bit-identical to auto: %s" % (t, sum(res[t]), bool(torch.equal(out, ref))), flush=True)
best = list(base)
pick = [0] * len(base)
for i in range(len(base)):
    for t, r in res.items():
        if r[i] < best[i]:
            best[i], pick[i] = r[i], t
print("best per launch: %.3f ms" % sum(best))
print("picks:", pick)
for i in range(len(base)):
    print(i, "%.4f" % base[i], " ".join("%.4f" % res[t][i] for t in (41, 42, 61, 62, 63, 64, 65, 66, 67)))
