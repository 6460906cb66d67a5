import sys, torch
sys.path.insert(0, '/root/repo')
from adafocus_amd import hip_ops as H, _lib as L
dev = torch.device('cuda:0')
torch.manual_seed(3)
for (hw, c, k) in ((5, 2304, 3), (5, 1392, 5), (4, 384, 3), (3, 1392, 5)):
    for dt in (torch.float32, torch.float16):
        x = (torch.randn(6, hw, hw, c, device=dev) * 0.7).to(dt)
        w = torch.randn(k * k, c, device=dev) * 0.3
        sc = torch.rand(c, device=dev) + 0.5
        bi = torch.randn(c, device=dev) * 0.1
        o1, p1 = H.dwconv_same_bn_act(x, w, sc, bi, k, 1, want_pool=True)
        with L.option("effnet_plan", 29):
            o2, p2 = H.dwconv_same_bn_act(x, w, sc, bi, k, 1, want_pool=True)
        print(hw, c, k, dt, torch.equal(o1, o2), torch.equal(p1, p2), int((p1 != p2).sum()), float((p1 - p2).abs().max()))
