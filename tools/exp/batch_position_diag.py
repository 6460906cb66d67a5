"""Does a frame's EfficientNet output depend on its position in the batch?  Images 1 and 3 of a batch of five against the same frames alone, after
blocks 3 ... 12, for several `effnet_plan` settings (the round-4 hunt for the single-rounding fp16 conversions).  usage: python tools/exp/batch_position_diag.py"""
import sys, torch
sys.path.insert(0, '/root/repo')
from adafocus_amd import synth, _lib as L
from adafocus_amd.efficientnet import EfficientNet
from adafocus_amd.utils import nchw_to_nhwc4
dev = torch.device('cuda:0')
torch.manual_seed(11)
m = EfficientNet.from_name('efficientnet-b3', num_classes=200, dtype='f16').eval()
shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1007).items()})
m = m.to(dev)
x = torch.randn(5, 3, 144, 144, device=dev) * 0.5
x4 = nchw_to_nhwc4(x)
with torch.no_grad():
    m.features_nhwc4(x4)
    net = m._net
    for plan in (63, 31, 31 - 8, 31 - 4, 31 - 16, 31 - 1, 0):
        with L.option("effnet_plan", plan):
            res = []
            for bi in (3, 4, 5, 6, 9, 12):
                a = net.forward_blocks(x4, bi).clone()
                cnt = []
                for i in (1, 3):
                    b = net.forward_blocks(x4[i:i + 1].contiguous(), bi).clone()
                    cnt.append(int((a[i:i + 1] != b).sum()))
                res.append((bi, cnt))
            print("plan", plan, res, flush=True)
