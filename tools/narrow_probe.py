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
n = 512
for name, hw, cin, cout, res in (("b1 proj 32->16 @112", 112, 32, 16, False), ("b2 proj 96->24 @56", 56, 96, 24, False), ("b3 proj 144->24 @56", 56, 144, 24, True),
                                 ("b4 proj 144->32 @28", 28, 144, 32, False), ("b5 proj 192->32 @28", 28, 192, 32, True),
                                 ("b11 proj 384->96 @14", 14, 384, 96, False), ("b12 proj 576->96 @14", 14, 576, 96, True),
                                 ("b14 proj 576->160 @7", 7, 576, 160, False), ("b15 proj 960->160 @7", 7, 960, 160, True),
                                 ("b8 expand 64->384 @14", 14, 64, 384, False), ("b12 expand 96->576 @14", 14, 96, 576, False)):
    x = torch.randn((n, hw, hw, cin), device=dev); w = torch.randn((cout, 1, 1, cin), device=dev) * 0.05
    sc, bi = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    r = torch.randn((n, hw, hw, cout), device=dev) if res else None
    out = torch.empty((n, hw, hw, cout), device=dev)
    row = [name]
    ref = None
    for tile in (0, 38, 39, 33, 32):
        ms = t(lambda: ops.conv2d_bn_act(x, w, sc, bi, r, 1, 0, ops.ACT_NONE, tile=tile, out=out))
        if ref is None: ref = out.clone()
        row.append("tile%d %.3f ms%s" % (tile, ms, "" if torch.equal(out, ref) else " (DIFF)"))
    print(" | ".join(row), flush=True)
