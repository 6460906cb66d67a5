#!/usr/bin/env python3
"""What does the temporal shift cost the MobileNetV2 glancer?  512 frames of 224^2 (64 clips x 8), with and without 8 segments, alternating."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adafocus_amd import synth  # noqa: E402
from adafocus_amd.mobilenet import mobilenet_v2  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
net = mobilenet_v2().eval()
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 3).items()})
net = net.to(dev)
eng = net._engine if hasattr(net, "_engine") else None
x4 = torch.randn((n, 224, 224, 4), device=dev)
x4[..., 3] = 0


def ms(seg, reps=10):
    for _ in range(3):
        eng.features(x4, tsm_segments=seg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        eng.features(x4, tsm_segments=seg)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


with torch.no_grad():
    net.features_from_nhwc4(x4)
    eng = net._engine
    res = {0: [], 8: []}
    for _ in range(4):
        for seg in (8, 0):
            res[seg].append(ms(seg))
for seg in (8, 0):
    print("segments=%d  %d frames: %s  min %.3f ms" % (seg, n, " ".join("%.3f" % v for v in res[seg]), min(res[seg])))
