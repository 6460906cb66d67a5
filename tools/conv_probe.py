#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM conv engine: every (shape, tile) pair timed with HIP
events on the launch stream.  Used to tune the tile heuristic and to profile single shapes
(rocprofv3 --pmc ... -- python tools/conv_probe.py --only NAME --tile T --iters N)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import hip_ops as ops  # noqa: E402


def resnet_shapes(n, p):
    """(name, n, h, w, cin, cout, k, stride, pad, residual) of every distinct conv of the trunk."""
    out = [("stem", n, p, p, 4, 64, 7, 2, 3, False)]
    hw = (p + 6 - 7) // 2 + 1
    hw = (hw + 2 - 3) // 2 + 1
    inpl = 64
    for s, (pl, nb, st) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)), start=1):
        for b in range(2):
            stride = st if b == 0 else 1
            tag = "L%d.%d" % (s, b)
            out.append((tag + ".c1", n, hw, hw, inpl, pl, 1, 1, 0, False))
            out.append((tag + ".c2", n, hw, hw, pl, pl, 3, stride, 1, False))
            ohw = (hw + 2 - 3) // stride + 1
            if b == 0:
                out.append((tag + ".ds", n, hw, hw, inpl, pl * 4, 1, stride, 0, False))
            out.append((tag + ".c3", n, ohw, ohw, pl, pl * 4, 1, 1, 0, True))
            hw, inpl = ohw, pl * 4
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--patch", type=int, default=96)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--tile", type=str, default="", help="comma list of tile ids; default = auto,1..4")
    ap.add_argument("--gemm", action="store_true", help="also time big plain GEMMs (asymptotic main-loop rate)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    shapes = resnet_shapes(a.n, a.patch)
    if a.gemm:
        shapes += [("gemm8k_k4096", 8192, 1, 1, 4096, 8192, 1, 1, 0, False), ("gemm64k_k512", 65536, 1, 1, 512, 1024, 1, 1, 0, False)]
    if a.only:
        shapes = [s for s in shapes if s[0] in a.only.split(",")]
    tiles = [int(t) for t in a.tile.split(',')] if a.tile else [0, 1, 2, 3, 4]
    print("%-14s %9s %6s | " % ("shape", "M", "K") + " ".join("%12s" % ("tile%d" % t) for t in tiles))
    for name, n, h, w, cin, cout, k, stride, pad, res in shapes:
        x = torch.randn((n, h, w, cin), device=dev)
        wt = torch.randn((cout, k, k, cin), device=dev) * 0.05
        sc = torch.rand(cout, device=dev) + 0.5
        bi = torch.randn(cout, device=dev) * 0.1
        oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        r = torch.randn((n, oh, ow, cout), device=dev) if res else None
        out = torch.empty((n, oh, ow, cout), device=dev)
        flops = 2.0 * n * oh * ow * cout * k * k * (3 if name == "stem" else cin)
        cells = []
        for t in tiles:
            for _ in range(2):
                ops.conv2d_bn_act(x, wt, sc, bi, r, stride, pad, ops.ACT_RELU, tile=t, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                ops.conv2d_bn_act(x, wt, sc, bi, r, stride, pad, ops.ACT_RELU, tile=t, out=out)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            cells.append("%6.3f/%5.1f" % (ms, flops / ms / 1e9))
        print("%-14s %9d %6d | " % (name, n * oh * ow, k * k * cin) + " ".join("%12s" % c for c in cells), flush=True)


if __name__ == "__main__":
    main()
