#!/usr/bin/env python3
"""Per-launch best tile by measurement: runs the trunk with every conv forced to each candidate tile and keeps,
per launch, the fastest.  Prints the totals for the native candidates and for native+split candidates.
usage: autotune_probe.py [patch=96] [patches=1024] [tsm=0]"""
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
trunk.set_fusion(False)        # one launch per layer, so the launch lists of all runs line up
nconv = 53


def run(tiles):
    trunk.set_tiles(tiles)
    for _ in range(2):
        trunk.forward(x, tsm_segments=tsm) if tsm else trunk.forward(x)
    runs = [trunk.profile(x, tsm_segments=tsm) for _ in range(3)]
    return [min(r[i]["ms"] for r in runs) for i in range(len(runs[0]))]


base = run([0] * nconv)
print("auto: %.3f ms" % sum(base))
res = {}
for t in (31, 32, 33, 34, 71, 72, 73, 74):
    res[t] = run([0] + [t] * (nconv - 1))
    print("all tile %d: %.3f ms" % (t, sum(res[t])), flush=True)
# launches: stem, maxpool, convs..., avgpool  -> conv index c maps to launch c+1 for c >= 1
best_native, best_any, pick_native, pick_any = list(base), list(base), [0] * len(base), [0] * len(base)
for i in range(len(base)):
    for t, r in res.items():
        if t < 40 and r[i] < best_native[i]:
            best_native[i], pick_native[i] = r[i], t
        if r[i] < best_any[i]:
            best_any[i], pick_any[i] = r[i], t
print("best native per launch: %.3f ms" % sum(best_native))
print("best native+split per launch: %.3f ms" % sum(best_any))
print("picks native:", pick_native)
print("picks any:", pick_any)
print("per-launch ms (launch: auto 31 32 33 34 41 42 43 44)")
for i in range(len(base)):
    print(i, "%.4f" % base[i], " ".join("%.4f" % res[t][i] for t in (31, 32, 33, 34, 71, 72, 73, 74)))
