#!/usr/bin/env python3
"""The GRU classifier alone (input-projection GEMM + persistent scan with the per-step FC folded in): ms per call for (B, T) cases.
usage: python tools/gru_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import _lib, hip_ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
F, H, C = 3328, 1024, 200
w_ih, w_hh = (torch.randn(3 * H, F, generator=g) * 0.02).to(dev), (torch.randn(3 * H, H, generator=g) * 0.03).to(dev)
b_ih, b_hh = torch.zeros(3 * H, device=dev), torch.zeros(3 * H, device=dev)
fc_w, fc_b = (torch.randn(C, H, generator=g) * 0.03).to(dev), torch.zeros(C, device=dev)
for b, t in ((64, 16), (64, 8), (32, 16), (2, 8), (1, 8), (200, 16)):
    x = torch.randn(b, t, F, generator=g).to(dev)
    for mode in (1, 2, 0):
        hip_ops.set_gru_persistent(1 if mode else 0, dev)
        _lib.set_option("gru_scan_slices", 2 if mode == 2 else 1)
        for _ in range(3):
            hip_ops.gru_cls_forward(x, w_ih, w_hh, b_ih, b_hh, fc_w, fc_b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            hip_ops.gru_cls_forward(x, w_ih, w_hh, b_ih, b_hh, fc_w, fc_b)
        e1.record()
        torch.cuda.synchronize()
        print("B=%3d T=%2d %s: %.3f ms per call" % (b, t, ("persistent scan, 1 slice " if mode == 1 else "persistent scan, 2 slices" if mode == 2 else "launch per step          "), e0.elapsed_time(e1) / 20))
hip_ops.set_gru_persistent(1, dev)
_lib.set_option("gru_scan_slices", 2)
