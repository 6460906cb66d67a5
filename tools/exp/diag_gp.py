import sys, torch
sys.path.insert(0, '/root/repo')
from adafocus_amd import hip_ops as H
dev = torch.device('cuda:0')
torch.manual_seed(3)
for (hw, k, n) in ((36, 192, 32), (18, 288, 48), (72, 144, 32), (72, 40, 24)):
    for dt in (torch.float32, torch.float16):
        x = (torch.randn(5, hw, hw, k, device=dev) * 0.7).to(dt)
        w = (torch.randn(n, k, device=dev) * 0.1).to(dt)
        g = torch.rand(5, k, device=dev)
        sc = torch.rand(n, device=dev) + 0.5
        bi = torch.randn(n, device=dev) * 0.1
        r = (torch.randn(5, hw, hw, n, device=dev)).to(dt)
        o5 = H.conv1x1_gated_bn(x, g, w, sc, bi, r)
        o5b = H.conv1x1_gated_bn(x, g, w, sc, bi, r)
        o1 = H.conv1x1_gated_bn(x[3:4].contiguous(), g[3:4].contiguous(), w, sc, bi, r[3:4].contiguous())
        print(hw, k, n, dt, torch.equal(o5, o5b), torch.equal(o5[3:4], o1), float((o5[3:4].float() - o1.float()).abs().max()))
    pm = torch.randn(5, k, device=dev)
    sq = max(k // 24, 8)
    wr, br, we, be = torch.randn(sq, k, device=dev) * 0.1, torch.randn(sq, device=dev) * 0.1, torch.randn(k, sq, device=dev) * 0.1, torch.randn(k, device=dev) * 0.1
    g5 = H.se_gate(pm, wr, br, we, be); g1 = H.se_gate(pm[3:4].contiguous(), wr, br, we, be)
    print("gate", k, torch.equal(g5[3:4], g1))
