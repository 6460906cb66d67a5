#!/usr/bin/env python3
"""EfficientNet-B3 per-block cost: forward truncated after k blocks (k = 0 is the stem), differences of consecutive times.
usage: python tools/effnet_blocks.py [N=1024] [P=144] [dtype=f16] [fusion=1]      (fusion=0: the four-launch plan for every block)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import synth  # noqa: E402
from adafocus_amd.efficientnet import EfficientNet  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
p = int(sys.argv[2]) if len(sys.argv) > 2 else 144
dt = sys.argv[3] if len(sys.argv) > 3 else "f16"
fusion = (int(sys.argv[4]) if len(sys.argv) > 4 else 1) != 0
dev = torch.device("cuda:0")
x4 = torch.randn((n, p, p, 4), device=dev)
x4[..., 3] = 0
m = EfficientNet.from_name("efficientnet-b3", num_classes=200, dtype=dt).eval()
shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1007).items()})
m = m.to(dev)
m.fusion = fusion
net = m.engine()
print("whole-block launches per forward: %d" % net.whole_blocks(p))
blocks = net.blocks()


def timeit(fn, iters=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


prev = 0.0
with torch.no_grad():
    for k in range(len(blocks) + 1):
        t = timeit(lambda: net.forward_blocks(x4, k))
        if k == 0:
            print("stem: %.3f ms" % t)
        else:
            b = blocks[k - 1]
            print("b%-2d k%d s%d cin %3d hid %4d cout %3d: %.3f ms (cumulative %.3f)" % (k - 1, b["k"], b["stride"], b["cin"], b["hid"], b["cout"], t - prev, t))
        prev = t
    full = timeit(lambda: m.features_nhwc4(x4))
    print("whole network: %.3f ms (head + pool: %.3f)" % (full, full - prev))
