#!/usr/bin/env python3
"""Randomised shapes through the conv engine (automatic tile choice and a few forced tiles) against the naive on-device
kernel of the same contract.  usage: conv_fuzz.py [cases=150] [seed=0]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import hip_ops as ops  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for i in range(cases):
    k = int(rng.choice([1, 1, 1, 3, 3, 5, 7]))
    stride = int(rng.choice([1, 1, 2])) if k > 1 or rng.random() < 0.3 else 1
    cin = int(rng.choice([4, 8, 16, 24, 32, 48, 64, 96, 128, 144, 160, 192, 256, 320, 384, 512]))
    cout = int(rng.choice([8, 16, 24, 32, 40, 64, 96, 128, 160, 200, 256, 320, 512, 1000]))
    if k >= 5:
        cin = min(cin, 32)
    hw = int(rng.integers(max(k, 2), 29))
    n = int(rng.integers(1, 40))
    if rng.random() < 0.25:
        n = int(rng.integers(200, 1200))
        hw = int(rng.integers(max(k, 2), 9))
    act = int(rng.choice([0, 1, 2, 3]))
    pad = k // 2 if rng.random() < 0.8 else 0
    oh = (hw + 2 * pad - k) // stride + 1
    if oh <= 0:
        continue
    res = rng.random() < 0.4
    tsm = 0
    if k == 1 and stride == 1 and pad == 0 and cin % 32 == 0 and rng.random() < 0.3:
        tsm = int(rng.choice([2, 4, 8]))
        n = max(tsm, n - n % tsm)
    x = torch.randn((n, hw, hw, cin), device=dev)
    w = torch.randn((cout, k, k, cin), device=dev) * float(1.0 / np.sqrt(k * k * cin))
    sc = torch.rand(cout, device=dev) + 0.5
    bi = torch.randn(cout, device=dev) * 0.1
    r = torch.randn((n, oh, oh, cout), device=dev) if res else None
    kw = dict(stride=stride, pad=pad, act=act, tsm_segments=tsm, tsm_div=8, residual=r)
    ref = ops.conv2d_bn_act(x, w, sc, bi, naive=True, **kw)
    for tile in (0, 33, 38, 3, 41):
        got = ops.conv2d_bn_act(x, w, sc, bi, tile=tile, **kw)
        err = (got - ref).abs().max().item()
        scale = max(ref.abs().max().item(), 1.0)
        worst = max(worst, err / scale)
        if not torch.isfinite(got).all() or err / scale > 2e-4:
            print("MISMATCH case %d tile %d: n=%d hw=%d cin=%d cout=%d k=%d s=%d pad=%d act=%d res=%s tsm=%d err=%.3e" %
                  (i, tile, n, hw, cin, cout, k, stride, pad, act, res, tsm, err), flush=True)
print("%d cases x 5 tiles done; worst relative error %.2e" % (cases, worst))
