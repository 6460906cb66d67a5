import sys, torch
sys.path.insert(0, '/root/repo')
from adafocus_amd import synth, _lib as L
from adafocus_amd.efficientnet import EfficientNet
from adafocus_amd.utils import nchw_to_nhwc4
dev = torch.device('cuda:0')
m = EfficientNet.from_name('efficientnet-b3', num_classes=200, dtype='f16', image_size=None).eval()
shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1007).items()})
m = m.to(dev)
x4 = nchw_to_nhwc4(torch.randn(5, 3, 75, 75, device=dev) * 0.5)
with torch.no_grad(), L.option('effnet_plan', 31):
    m.features_nhwc4(x4)
    torch.cuda.synchronize()
