#!/usr/bin/env python3
"""One sha256 per tiny-map depthwise case (output map; the squeeze sums are reported rounded: their summation order differs
between the two kernels).  tests/test_effnet.py calls digests() with the library option "effnet_plan" at its default (one thread
per image x 4 channels, padding taps skipped: csrc/effnet.hip dw_small_kernel) and with ADAF_EF_PLAN_TINY_DW cleared
(dw_same_kernel) and expects the same digests.  Run as a script it prints both arms."""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import _lib, hip_ops as ops  # noqa: E402


def digests():
    dev = torch.device("cuda:0")
    out_rows = []
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
                    out_rows.append((str(i), hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest(), "%.4f" % float(pool.double().sum())))
                    i += 1
    return out_rows


if __name__ == "__main__":
    plan = int(_lib.get_option("effnet_plan"))
    for label, value in (("tiny-map kernel", plan | _lib.EF_PLAN_TINY_DW), ("staged kernel", plan & ~_lib.EF_PLAN_TINY_DW)):
        with _lib.option("effnet_plan", value):
            for row in digests():
                print(label, *row)
