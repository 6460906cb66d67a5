"""A/B of one bit of the library option "effnet_plan" on the EfficientNet-B3 local CNN (fp16 storage): N patches of P^2, the two plans
timed alternately.  usage: python tools/effnet_plan_ab.py BIT [N=1024] [P=144] [iters=20] [rounds=3]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import _lib, synth  # noqa: E402
from adafocus_amd.efficientnet import EfficientNet  # noqa: E402

bit = int(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
p = int(sys.argv[3]) if len(sys.argv) > 3 else 144
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dev = torch.device("cuda:0")
x4 = torch.randn((n, p, p, 4), device=dev)
x4[..., 3] = 0
m = EfficientNet.from_name("efficientnet-b3", num_classes=200, dtype="f16").eval()
shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1007).items()})
m = m.to(dev)
plan = int(_lib.get_option("effnet_plan"))


def timed(value):
    with _lib.option("effnet_plan", value), torch.no_grad():
        for _ in range(3):
            out = m.features_nhwc4(x4)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            out = m.features_nhwc4(x4)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, out.clone()


for r in range(rounds):
    on, a = timed(plan | bit)
    off, b = timed(plan & ~bit)
    print("round %d: bit %d on %.3f ms, off %.3f ms (%d x %d^2, fp16 storage); equal features: %s, max |d| %.3e"
          % (r, bit, on, off, n, p, torch.equal(a, b), (a - b).abs().max().item()))
