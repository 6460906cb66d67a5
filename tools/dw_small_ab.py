#!/usr/bin/env python3
"""Print one sha256 per tiny-map depthwise case (output map; the squeeze means are printed rounded, their summation order differs
between the two kernels).  tests/test_effnet.py runs it with ADAF_DW_SMALL=1 (one thread per image x 4 channels, padding taps
skipped: csrc/effnet.hip dw_small_kernel, the default) and =0 (dw_same_kernel) and expects the same digests."""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import hip_ops as ops  # noqa: E402

dev = torch.device("cuda:0")
i = 0
for hw in (3, 4, 5):
    for k in (3, 5):
        for dt in (torch.float32, torch.float16):
            for n, c in ((7, 48), (3, 1392)):
                g = np.random.Generator(np.random.PCG64([i, 29]))
                x = torch.from_numpy(g.standard_normal((n, hw, hw, c), dtype=np.float32)).to(dev).to(dt)
                w = torch.from_numpy(g.standard_normal((k * k, c), dtype=np.float32) * np.float32(0.3)).to(dev)
                sc = torch.from_numpy(g.uniform(0.5, 1.5, c).astype(np.float32)).to(dev)
                bi = torch.from_numpy(g.normal(0, 0.1, c).astype(np.float32)).to(dev)
                out, pool = ops.dwconv_same_bn_act(x, w, sc, bi, k, 1, ops.ACT_SWISH, want_pool=True)
                print(i, hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest(), "%.4f" % float(pool.double().sum()))
                i += 1
