#!/usr/bin/env python3
"""Kernel-level view of the glancer: run N forwards of adaf_mobilenetv2 at the bench shape (under rocprofv3)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import synth  # noqa: E402
from adafocus_amd.mobilenet import mobilenet_v2  # noqa: E402

dev = torch.device("cuda:0")
net = mobilenet_v2().eval()
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 3).items()})
net = net.to(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
if len(sys.argv) > 2:       # adaf_mobilenetv2_set_fusion bits: 1 fused kernels, 4 no chunk pairing
    net._engine.fusion = int(sys.argv[2])
x4 = torch.randn((n, 224, 224, 4), device=dev)
x4[..., 3] = 0
with torch.no_grad():
    for _ in range(2):
        net.features_from_nhwc4(x4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        net.features_from_nhwc4(x4)
    e1.record()
    torch.cuda.synchronize()
print("glancer (from pixel-major frames) %d frames: %.3f ms" % (n, e0.elapsed_time(e1) / 3))
