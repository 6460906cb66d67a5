import sys, torch
sys.path.insert(0, '/root/repo')
from adafocus_amd import hip_ops as H
dev = torch.device('cuda:0')
torch.manual_seed(3)
for (hw, c, k, s) in ((36, 192, 3, 1), (72, 144, 3, 2), (72, 40, 3, 1), (36, 192, 5, 2), (18, 288, 5, 1)):
    for dt in (torch.float32, torch.float16):
        x = (torch.randn(5, hw, hw, c, device=dev) * 0.7).to(dt)
        w = torch.randn(k * k, c, device=dev) * 0.3
        sc = torch.rand(c, device=dev) + 0.5
        bi = torch.randn(c, device=dev) * 0.1
        o5, p5 = H.dwconv_same_bn_act(x, w, sc, bi, k, s, want_pool=True)
        o1, p1 = H.dwconv_same_bn_act(x[3:4].contiguous(), w, sc, bi, k, s, want_pool=True)
        print(hw, c, k, s, dt, torch.equal(o5[3:4], o1), torch.equal(p5[3:4], p1), float((p5[3:4] - p1).abs().max()))
