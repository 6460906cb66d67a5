#!/usr/bin/env python3
"""Accuracy of the split-bf16 tiles (4x = 6 products, 5x = 9 products) against an fp64 convolution, next to
the native fp32-MFMA tile and torch's CPU fp32 conv.  Error metric: max |y - y64| / rms(y64)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import hip_ops as ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
cases = [("3x3 256->256 @6x6", 64, 6, 6, 256, 256, 3, 1, 1), ("1x1 1024->256 @6x6", 64, 6, 6, 1024, 256, 1, 1, 0),
         ("3x3 512->512 @3x3", 64, 3, 3, 512, 512, 3, 1, 1), ("1x1 64->256 @24x24", 8, 24, 24, 64, 256, 1, 1, 0)]
for name, n, h, w, cin, cout, k, st, pad in cases:
    x = torch.randn((n, h, w, cin)).abs_()          # post-ReLU-like, non-negative: no cancellation hides errors
    x *= torch.exp(torch.randn((n, h, w, cin)))     # wide dynamic range
    wt = torch.randn((cout, k, k, cin)) * (1.0 / (k * k * cin) ** 0.5)
    sc = torch.ones(cout)
    bi = torch.zeros(cout)
    y64 = F.conv2d(x.permute(0, 3, 1, 2).double(), wt.permute(0, 3, 1, 2).double(), None, st, pad).permute(0, 2, 3, 1)
    y32 = F.conv2d(x.permute(0, 3, 1, 2), wt.permute(0, 3, 1, 2), None, st, pad).permute(0, 2, 3, 1)
    rms = y64.pow(2).mean().sqrt().item()
    row = ["%-20s rms %.3g | torch-cpu-f32 %.2e" % (name, rms, (y32.double() - y64).abs().max().item() / rms)]
    for t in (33, 43, 53):
        y = ops.conv2d_bn_act(x.to(dev), wt.to(dev), sc.to(dev), bi.to(dev), None, st, pad, ops.ACT_NONE, tile=t)
        e = (y.cpu().double() - y64)
        row.append("tile%d max %.2e rms %.2e" % (t, e.abs().max().item() / rms, e.pow(2).mean().sqrt().item() / rms))
    print(" | ".join(row), flush=True)
