#!/usr/bin/env python3
"""Round 6: the strip-walking front kernels (csrc/mbstrip.hip, option `mb_strip`) against the wave-private tiles they replace:
torch.equal of the glancer's outputs at several frame sizes, then the 1024-frame glancer timed with the option on / off, alternating.
usage: strip_ab.py [frames=1024] [rounds=4] [fusion bits]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adafocus_amd import _lib, synth  # noqa: E402
from adafocus_amd.mobilenet import mobilenet_v2  # noqa: E402

dev = torch.device("cuda:0")
net = mobilenet_v2().eval()
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 3).items()})
net = net.to(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
if len(sys.argv) > 3:
    net._engine.fusion = int(sys.argv[3])


def fwd(x4, strip):
    _lib.set_option("mb_strip", strip)
    with torch.no_grad():
        fm, fv = net.features_from_nhwc4(x4)
    return fm.clone(), fv.clone()


ok = True
for nf, size in ((5, 224), (3, 56), (2, 84), (3, 140), (3, 200), (520, 56)):
    g = torch.Generator(device="cpu").manual_seed(size + nf)
    x4 = torch.zeros((nf, size, size, 4))
    x4[..., :3] = torch.randn((nf, size, size, 3), generator=g)
    x4 = x4.to(dev)
    a, av = fwd(x4, 1)
    b, bv = fwd(x4, 0)
    same = torch.equal(a, b) and torch.equal(av, bv)
    ok &= same
    print("equal(strip, tiles) n=%d size=%d: %s  max|d| %.3e  (scale %.2f)" % (nf, size, same, (a - b).abs().max().item(), b.abs().max().item()), flush=True)
print("ALL EQUAL" if ok else "MISMATCH")
# run-to-run: the same 512-frame batch ten times (a load landing in an in-flight MFMA's register shows up here, not against a tolerance)
xr = torch.randn((512, 224, 224, 4), device=dev)
xr[..., 3] = 0
ref = fwd(xr, 1)
same = all(torch.equal(ref[0], fwd(xr, 1)[0]) for _ in range(10))
print("run-to-run identical (10 x 512 frames): %s" % same, flush=True)

x4 = torch.randn((n, 224, 224, 4), device=dev)
x4[..., 3] = 0


def timed(strip, iters=3):
    _lib.set_option("mb_strip", strip)
    with torch.no_grad():
        net.features_from_nhwc4(x4)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            net.features_from_nhwc4(x4)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


timed(1), timed(0)
for r in range(rounds):
    print("round %d: glancer %d frames  strip %.3f ms   tiles %.3f ms" % (r, n, timed(1), timed(0)), flush=True)
_lib.set_option("mb_strip", 1)
