#!/usr/bin/env python3
"""Split-bf16 trunk: per-launch time under each pre-split-weight tile (61-67) against the automatic pick.
usage: split_tiles_probe.py [patch=96] [patches=1024]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import synth  # noqa: E402
from adafocus_amd.resnet import resnet50  # noqa: E402

dev = torch.device("cuda:0")
p = int(sys.argv[1]) if len(sys.argv) > 1 else 96
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
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
        trunk.forward(x)
    runs = [trunk.profile(x) for _ in range(3)]
    return [min(r[i]["ms"] for r in runs) for i in range(len(runs[0]))], [r["tile"] for r in runs[0]]


base, used = run([0] * nconv)
print("auto: %.3f ms" % sum(base))
res = {}
cands = (61, 62, 63, 64, 65, 66, 67)
for t in cands:
    res[t], _ = run([0] + [t] * (nconv - 1))
    print("all tile %d: %.3f ms" % (t, sum(res[t])), flush=True)
best = list(base)
pick = list(used)
for i in range(len(base)):
    for t, r in res.items():
        if len(r) == len(base) and r[i] < best[i] * 0.98:
            best[i], pick[i] = r[i], t
print("best per launch: %.3f ms" % sum(best))
print("launch: auto(tile) | " + " ".join(str(t) for t in cands) + " | pick")
for i in range(len(base)):
    print(i, "%.4f(%d)" % (base[i], used[i]), " ".join("%.4f" % res[t][i] if len(res[t]) == len(base) else "-" for t in cands), "|", pick[i])
