import sys, torch
sys.path.insert(0, '/root/repo')
from adafocus_amd import synth, _lib as L
from adafocus_amd.efficientnet import EfficientNet
dev = torch.device('cuda:0')
m = EfficientNet.from_name('efficientnet-b3', num_classes=200, dtype='f16').eval()
shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1007).items()})
m = m.to(dev)
xb = torch.randn((1024, 144, 144, 4), device=dev); xb[..., 3] = 0
L.set_option(sys.argv[1], float(sys.argv[2]))
with torch.no_grad():
    for _ in range(8): m.features_nhwc4(xb)
    torch.cuda.synchronize()
