#!/usr/bin/env python3
"""Per-launch table of the ResNet-50 trunk (HIP-event bracketed): ms, TFLOP/s, algorithmic TB/s, tile.
usage: layer_table.py [patch=96] [patches=1024] [tsm_segments=0] [fusion=1]"""
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
if os.environ.get("MATH"):          # MATH=split_bf16: the opt-in arithmetic
    net.set_math(os.environ["MATH"])
trunk = net._sync()
for _ in range(3):
    trunk.forward(x, tsm_segments=tsm) if tsm else trunk.forward(x)
runs = [trunk.profile(x, tsm_segments=tsm) for _ in range(5)]
fuse = int(sys.argv[4]) if len(sys.argv) > 4 else 1
if len(sys.argv) > 5:      # 0 = row-major tiles, 1 = position-major with tap skipping (default), 2 = position-major without skipping
    from adafocus_amd import hip_ops
    hip_ops.set_conv_pos_major(int(sys.argv[5]), dev)
trunk.set_fusion(fuse)
for _ in range(2):
    trunk.forward(x, tsm_segments=tsm) if tsm else trunk.forward(x)
runs = [trunk.profile(x, tsm_segments=tsm) for _ in range(5)]
# launch names follow the plan: tile 90 = stem + max-pool in one launch, 91 = conv2 -> conv3, 92 = conv2 -> conv3 -> next conv1,
# 93 = layer1.0's conv1 + downsample conv in one launch
order = []
for li, nb in enumerate((3, 4, 6, 3), 1):
    for b in range(nb):
        order += [("L%d.%d" % (li, b), b == 0)]
names, skip_c1 = [], False
it = iter(runs[0])
e = next(it)
names.append("stem+pool" if e["tile"] == 90 else "stem")
if e["tile"] != 90:
    next(it)
    names.append("maxpool")
for blk, has_ds in order:
    merged = False
    if not skip_c1:
        e = next(it)
        merged = e["tile"] == 93          # layer1.0: conv1 and the downsample conv in one launch
        names.append(blk + (".c1+ds" if merged else ".c1"))
    skip_c1 = False
    if has_ds and not merged:
        next(it)
        names.append(blk + ".ds")
    e = next(it)
    if e["tile"] in (91, 92):
        names.append(blk + (".c2+c3+c1'" if e["tile"] == 92 else ".c2+c3"))
        skip_c1 = e["tile"] == 92
    else:
        names.append(blk + ".c2")
        next(it)
        names.append(blk + ".c3")
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
