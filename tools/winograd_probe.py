import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import hip_ops as ops
dev = torch.device("cuda:0")
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
# Winograd F(2x2,3x3) GEMM stage for the trunk's stride-1 3x3 layers at 1024 patches of 96^2
for name, tiles_per_img, c in (("L1 (24x24, 64ch)", 144, 64), ("L2 (12x12, 128ch)", 36, 128), ("L3 (6x6, 256ch)", 9, 256), ("L4 (3x3, 512ch)", 4, 512)):
    m = 16 * tiles_per_img * 1024
    x = torch.randn((m, 1, 1, c), device=dev); w = torch.randn((c, 1, 1, c), device=dev) * 0.05
    sc, bi = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    out = torch.empty((m, 1, 1, c), device=dev)
    ms = t(lambda: ops.conv2d_bn_act(x, w, sc, bi, None, 1, 0, ops.ACT_NONE, out=out))
    direct_flops = 2.0 * (m / 16 * 4) * c * c * 9     # 4 outputs per tile
    mem_ms = (m * c * 4 * 2 * 1.0 + (m / 16 * 4) * c * 4 * 2) / 4.5e12 * 1e3   # V write+read, M write+read, in/out
    print("%-20s GEMM stage %.3f ms (%.0f TF on its own flops); transforms >= %.3f ms at 4.5 TB/s; direct-conv flops %.1f G"
          % (name, ms, 2.0 * m * c * c / ms / 1e9, mem_ms, direct_flops / 1e9))
