import sys, torch
sys.path.insert(0, '/root/repo')
from adafocus_amd import synth
from adafocus_amd.efficientnet import EfficientNet
from adafocus_amd.utils import nchw_to_nhwc4
dev = torch.device('cuda:0')
torch.manual_seed(11)
m = EfficientNet.from_name('efficientnet-b3', num_classes=200, dtype='f16').eval()
shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1007).items()})
m = m.to(dev)
for trial in range(3):
    x = torch.randn(5, 3, 144, 144, device=dev) * 0.5
    x4 = nchw_to_nhwc4(x)
    with torch.no_grad():
        m.features_nhwc4(x4)
        net = m._net
        for bi in range(1, 7):
            a = net.forward_blocks(x4, bi).clone()
            outs = []
            for i in range(5):
                b = net.forward_blocks(x4[i:i + 1].contiguous(), bi).clone()
                b2 = net.forward_blocks(x4[i:i + 1].contiguous(), bi).clone()
                d = (a[i:i + 1].float() - b.float()).abs()
                nz = (d > 0).nonzero()
                outs.append((int((d > 0).sum()), torch.equal(b, b2), nz[:3].tolist() if len(nz) else []))
            rep = net.forward_blocks(x4[[3, 3, 3, 3, 3]].contiguous(), bi)
            print(trial, bi, tuple(a.shape), outs, "rep-equal", [torch.equal(rep[i], rep[0]) for i in range(5)], flush=True)
