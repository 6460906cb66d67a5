#!/usr/bin/env python3
"""One sha256 per conv case (outputs of the default tile choice and three forced tiles).  tests/test_hip_parity_r2.py calls
digests() with the library option "conv_lean" = 1 (scalar-base DMA K loop + lean epilogue, the default) and = 0 (builtin DMA,
general epilogue) and expects the same digests: the lean forms change instructions, not arithmetic."""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import _lib, hip_ops as ops  # noqa: E402

CASES = [
    # n, h, w, cin, cout, k, stride, pad, act, residual
    (8, 12, 12, 256, 256, 1, 1, 0, 1, True),      # interior 128x64 tiles, 1x1 with identity
    (9, 12, 12, 64, 256, 1, 1, 0, 0, False),      # K = 64 (two slices), ragged last row tile
    (8, 12, 12, 512, 200, 1, 1, 0, 1, True),      # ragged columns: both epilogues in one launch
    (130, 3, 3, 256, 256, 3, 1, 1, 1, False),     # position-major tiles (130 images: one full group + a ragged one)
    (136, 6, 6, 128, 128, 3, 1, 1, 2, True),      # position-major, ReLU6 + identity
    (70, 6, 6, 128, 128, 3, 2, 1, 1, False),      # position-major, stride 2
    (3, 24, 24, 64, 64, 3, 1, 1, 1, False),       # row-major k x k (too few images for position-major tiles)
    (1, 5, 5, 96, 36, 1, 1, 0, 0, False),         # smaller than one tile
    (9, 12, 12, 256, 512, 1, 2, 0, 0, False),     # 1x1 / stride 2 (downsample branch): the lean dense kernel with a row gather
]


def digests():
    dev = torch.device("cuda:0")
    rows = []
    for i, (n, h, w, cin, cout, k, s, pad, act, res) in enumerate(CASES):
        g = np.random.Generator(np.random.PCG64([i, 23]))
        x = torch.from_numpy(g.standard_normal((n, h, w, cin), dtype=np.float32)).to(dev)
        wt = torch.from_numpy(g.standard_normal((cout, cin, k, k), dtype=np.float32) * np.float32(np.sqrt(2.0 / (cin * k * k)))).to(dev)
        sc = torch.from_numpy(g.uniform(0.5, 1.5, cout).astype(np.float32)).to(dev)
        bi = torch.from_numpy(g.normal(0, 0.1, cout).astype(np.float32)).to(dev)
        oh, ow = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
        r = torch.from_numpy(g.standard_normal((n, oh, ow, cout), dtype=np.float32)).to(dev) if res else None
        wp = ops.pack_conv_weight(wt, cin)
        for tile in (0, 31, 32, 33):
            y = ops.conv2d_bn_act(x, wp, sc, bi, stride=s, pad=pad, act=act, residual=r, tile=tile).cpu().numpy()
            rows.append((i, tile, hashlib.sha256(y.tobytes()).hexdigest()))
    return rows


if __name__ == "__main__":
    for value in (1, 0):
        with _lib.option("conv_lean", value):
            for row in digests():
                print("conv_lean=%d" % value, *row)
