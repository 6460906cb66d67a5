#!/usr/bin/env python3
"""EfficientNet-B3 at 144^2, the launches left on the conv engine / the gated project with fp16 storage (1024 patches, 5 x 5 maps):
block 25's expand (384 -> 2304, swish, fp16 out), its gated project (2304 -> 384), the head (384 -> 1536, swish, fp32 out): every fp16 tile id
against the automatic choice."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import hip_ops as ops
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
hw = 5
for name, k, cout, odt in (("b25 expand", 384, 2304, torch.float16), ("head", 384, 1536, torch.float32), ("b19 expand", 232, 1392, torch.float16)):
    x = (torch.randn((n, hw, hw, k), device=dev) * 0.5).half(); w = (torch.randn((cout, 1, 1, k), device=dev) * 0.05).half()
    sc, bi = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    fl = 2.0 * n * hw * hw * k * cout
    row = []
    for t in (0, 81, 82, 83, 84, 88):
        try:
            ms = timeit(lambda: ops.conv2d_bn_act_f16(x, w, sc, bi, None, 1, 0, ops.ACT_SWISH, out_dtype=odt, tile=t))
            row.append((ms, t))
        except Exception as e:
            print(name, t, "failed:", str(e)[:80])
    print("%-12s K=%4d N=%4d: " % (name, k, cout) + " ".join("%d:%.0f us (%.0f TF)" % (t, m * 1e3, fl / m / 1e9) for m, t in row))
k, cout = 2304, 384
x = (torch.randn((n, hw, hw, k), device=dev) * 0.5).half(); w = (torch.randn((cout, 1, 1, k), device=dev) * 0.02).half()
gate = torch.rand((n, k), device=dev)
sc, bi = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
ms = timeit(lambda: ops.conv1x1_gated_bn(x, gate, w, sc, bi, None))
print("b25 gated project K=%d N=%d: %.0f us (%.0f TF, %.2f TB/s of D)" % (k, cout, ms * 1e3, 2.0 * n * 25 * k * cout / ms / 1e9, n * 25 * k * 2 / ms / 1e9))
ms = timeit(lambda: ops.conv2d_bn_act_f16(x, w, sc, bi, None, 1, 0, ops.ACT_NONE, out_dtype=torch.float16, tile=0))
print("   the same GEMM without the gate on the engine (auto tile): %.0f us" % (ms * 1e3))
for t in (81, 82, 83, 84):
    ms = timeit(lambda: ops.conv2d_bn_act_f16(x, w, sc, bi, None, 1, 0, ops.ACT_NONE, out_dtype=torch.float16, tile=t))
    print("   tile %d: %.0f us" % (t, ms * 1e3))
