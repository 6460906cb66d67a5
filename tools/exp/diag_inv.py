import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from adafocus_amd import synth
from adafocus_amd.efficientnet import EfficientNet
from adafocus_amd.utils import nchw_to_nhwc4
dev = torch.device('cuda:0')
for dt in ('f32', 'f16'):
    m = EfficientNet.from_name('efficientnet-b3', num_classes=200, dtype=dt).eval()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1007).items()})
    m = m.to(dev)
    x = torch.randn(5, 3, 144, 144, device=dev) * 0.5
    x4 = nchw_to_nhwc4(x)
    with torch.no_grad():
        m.features_nhwc4(x4)
        net = m._net
        for bi in range(3, 7):
            a = net.forward_blocks(x4, bi).clone()
            a2 = net.forward_blocks(x4, bi).clone()
            b = net.forward_blocks(x4[3:4].contiguous(), bi).clone()
            print(dt, bi, tuple(a.shape), torch.equal(a, a2), torch.equal(a[3:4], b), float((a[3:4].float() - b.float()).abs().max()))
