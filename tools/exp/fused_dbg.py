import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from adafocus_amd import synth, _lib as L
from adafocus_amd.efficientnet import EfficientNet
from adafocus_amd.utils import nchw_to_nhwc4
dev = torch.device('cuda:0')
for size, isz in ((75, None), (75, "native"), (100, None)):
    m = EfficientNet.from_name('efficientnet-b3', num_classes=200, dtype='f16', image_size=isz).eval()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1007).items()})
    m = m.to(dev)
    x4 = nchw_to_nhwc4(torch.randn(5, 3, size, size, device=dev) * 0.5)
    with torch.no_grad():
        eng = m.engine()
        for k in range(2, 10):
            a = eng.forward_blocks(x4, k).float().clone()
            with L.option("effnet_plan", 31):
                b = eng.forward_blocks(x4, k).float().clone()
            print(size, isz, k, tuple(a.shape), int((a != b).sum()), float((a - b).abs().max()))
