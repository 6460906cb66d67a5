"""Stem + max-pool launch of the ResNet-50 trunk: the strip-walking kernel (option stem_rows = 1) against the tile form (0), per patch
size: time of the first launch of a profiled trunk pass, whole trunk back to back, and bit-identity of the features."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import _lib, synth  # noqa: E402
from adafocus_amd.resnet import resnet50  # noqa: E402

dev = torch.device("cuda:0")
net = resnet50(num_classes=200).eval()
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1007).items()})
net = net.to(dev)
n = int(os.environ.get("N", "1024"))
for p in (96, 128, 144, 64):
    x = torch.randn((n, p, p, 4), device=dev)
    x[..., 3] = 0
    res = {}
    for mode in (0, 1):
        with _lib.option("stem_rows", mode), torch.no_grad():
            trunk = net._sync()
            trunk.set_fusion(2)          # the fused stem at every size
            f = net.features_nhwc4(x).clone()
            trunk.profile(x)
            ms = min(trunk.profile(x)[0]["ms"] for _ in range(5))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                trunk.forward(x)
            e1.record()
            torch.cuda.synchronize()
            res[mode] = (f, ms, e0.elapsed_time(e1) / 3)
    print("P=%d n=%d: stem+pool %.3f ms (tiles) -> %.3f ms (strips); trunk %.3f -> %.3f ms; bit-identical %s"
          % (p, n, res[0][1], res[1][1], res[0][2], res[1][2], torch.equal(res[0][0], res[1][0])), flush=True)

