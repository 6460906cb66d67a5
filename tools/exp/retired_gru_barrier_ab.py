import os, sys, torch
sys.path.insert(0, "/root/repo")
from adafocus_amd import _lib, hip_ops
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
F, H, C = 3328, 1024, 200
w_ih, w_hh = (torch.randn(3 * H, F, generator=g) * 0.02).to(dev), (torch.randn(3 * H, H, generator=g) * 0.03).to(dev)
b_ih, b_hh = torch.zeros(3 * H, device=dev), torch.zeros(3 * H, device=dev)
fc_w, fc_b = (torch.randn(C, H, generator=g) * 0.03).to(dev), torch.zeros(C, device=dev)
for b, t in ((64, 16), (1, 8), (2, 8), (200, 16)):
    x = torch.randn(b, t, F, generator=g).to(dev)
    for bar in (0, 1, 0, 1):
        _lib.set_option("gru_barrier", bar)
        for _ in range(3):
            hip_ops.gru_cls_forward(x, w_ih, w_hh, b_ih, b_hh, fc_w, fc_b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            hip_ops.gru_cls_forward(x, w_ih, w_hh, b_ih, b_hh, fc_w, fc_b)
        e1.record()
        torch.cuda.synchronize()
        print("B=%3d T=%2d barrier %s: %.4f ms per call" % (b, t, "xcd " if bar else "flat", e0.elapsed_time(e1) / 50), flush=True)
