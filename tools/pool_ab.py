#!/usr/bin/env python3
"""One sha256 per trunk-feature case.  tests/test_hip_parity_r3.py calls digests() with the library option "conv_pool" = 1 (the last
conv3 averages its map in its epilogue: csrc/conv_gemm.hip conv_epilogue_pool, the default) and = 0 (conv3 + avgpool_kernel) and
expects the same digests."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import _lib, synth  # noqa: E402
from adafocus_amd.resnet import resnet50  # noqa: E402


def digests():
    dev = torch.device("cuda:0")
    net = resnet50(num_classes=200).eval()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1).items()})
    net = net.to(dev)
    t = net._sync()
    rows = []
    for p, n, tsm in ((96, 512, 0), (96, 200, 0), (128, 120, 0), (144, 70, 0), (128, 128, 8)):
        g = torch.Generator().manual_seed(p + n)
        x = torch.randn((n, p, p, 4), generator=g).to(dev)
        x[..., 3] = 0
        f = t.forward(x, tsm_segments=tsm) if tsm else t.forward(x)
        rows.append((p, n, tsm, hashlib.sha256(f.cpu().numpy().tobytes()).hexdigest()))
    return rows


if __name__ == "__main__":
    for value in (1, 0):
        with _lib.option("conv_pool", value):
            for row in digests():
                print("conv_pool=%d" % value, *row)
