"""Hot path with the gather folded into the stem (adaf_resnet50_forward_frames) against gather + trunk: ms per 1024 patches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import synth  # noqa: E402
from adafocus_amd.resnet import resnet50  # noqa: E402
from adafocus_amd.utils import get_patch_nhwc4  # noqa: E402

dev = torch.device("cuda:0")
net = resnet50(num_classes=200).eval()
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1007).items()})
net = net.to(dev)
n = 1024
frames = torch.randn((n, 3, 224, 224), device=dev)
f4 = torch.cat([frames, torch.zeros_like(frames[:, :1])], 1).permute(0, 2, 3, 1).contiguous()
act = torch.rand((n, 2), device=dev)


def t(fn, it=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


with torch.no_grad():
    for p in (96, 128, 144):
        a = t(lambda: net.features_nhwc4(get_patch_nhwc4(frames, act, p)))
        b = t(lambda: net.features_from_frames(frames, act, p))
        c = t(lambda: net.features_from_frames(f4, act, p))
        print("P=%d: gather + trunk %.3f ms; folded (planar frames) %.3f ms; folded (pixel-major frames) %.3f ms" % (p, a, b, c), flush=True)
