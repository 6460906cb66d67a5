#!/usr/bin/env python3
"""Times the ResNet-50 trunk for several batch sizes (patches per launch) to see where the
per-patch cost changes (cache residency of stage-1 activations vs tile quantisation)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import _lib, synth  # noqa: E402
from adafocus_amd.resnet import resnet50  # noqa: E402

dev = torch.device("cuda:0")
print("CUs reported:", _lib.load_library().adaf_device_cus(_lib.handle(dev)))
net = resnet50(num_classes=200).eval()
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1).items()})
net = net.to(dev)
p = int(sys.argv[1]) if len(sys.argv) > 1 else 96
for n in (64, 128, 256, 512, 1024, 2048):
    x = torch.randn((n, p, p, 4), device=dev)
    x[..., 3] = 0
    trunk = net._sync()
    for _ in range(2):
        trunk.forward(x)
    prof = trunk.profile(x)
    stage = [0.0] * 6
    names = ["stem+pool", "layer1", "layer2", "layer3", "layer4", "avgpool"]
    bounds = [2, 2 + 10, 2 + 10 + 13, 2 + 10 + 13 + 19, 2 + 10 + 13 + 19 + 10, 55]
    j = 0
    for i, e in enumerate(prof):
        while i >= bounds[j]:
            j += 1
        stage[j] += e["ms"]
    tot = sum(stage)
    print("N=%5d total %.3f ms  %.2f us/patch | " % (n, tot, 1e3 * tot / n) +
          "  ".join("%s %.2f" % (nm, 1e3 * s / n) for nm, s in zip(names, stage)), flush=True)
