#!/usr/bin/env python3
"""Phase timeline of the depthwise launches of EfficientNet-B3 (csrc/effnet.hip dw_same_kernel built with -DEF_TRACE by
tools/exp/build_mbw_trace.sh): s_memtime stamps of every wave at the phase boundaries, averaged over the workgroups of the launch of block K.
usage: python tools/dw_trace.py [block=2] [N=1024] [P=144]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("ADAF_LIB", os.path.join(ROOT, "adafocus_amd", "csrc", "libadafocus_hip_mbwtrace.so"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from adafocus_amd import _lib, synth  # noqa: E402
from adafocus_amd.efficientnet import EfficientNet  # noqa: E402

blk = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
p = int(sys.argv[3]) if len(sys.argv) > 3 else 144
dev = torch.device("cuda:0")
x4 = torch.randn((n, p, p, 4), device=dev)
x4[..., 3] = 0
m = EfficientNet.from_name("efficientnet-b3", num_classes=200, dtype="f16").eval()
shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1007).items()})
m = m.to(dev)
net = m.engine()
b = net.blocks()[blk]
print("block %d: k%d s%d cin %d hid %d cout %d" % (blk, b["k"], b["stride"], b["cin"], b["hid"], b["cout"]))
_lib.load_library()
raw = C.CDLL(_lib.LIB_PATH)
raw.adaf_ef_set_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
NB = 1 << 18
buf = torch.zeros((NB, 4, 8), dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(3):
        net.forward_blocks(x4, blk + 1)
    torch.cuda.synchronize()
    raw.adaf_ef_set_trace(buf.data_ptr(), b["hid"], b["k"], b["stride"])
    net.forward_blocks(x4, blk + 1)
    torch.cuda.synchronize()
    raw.adaf_ef_set_trace(None, 0, 0, 0)
t = buf.cpu().numpy().astype(np.float64)
live = t[:, 0, 0] > 0
t = t[live]
print("workgroups traced: %d" % len(t))
t0 = t[:, :, 0].min(axis=1)[:, None, None]
rel = np.where(t > 0, t - t0, np.nan)
names = ["start", "taps staged, filter rows in registers", "tile staged / expanded", "barrier", "taps done", "end (squeeze partials written)"]
print("%-40s %10s   per wave" % ("stamp", "mean"))
for s in range(6):
    col = rel[:, :, s]
    if np.all(np.isnan(col)):
        continue
    print("%-40s %10.0f   %s" % (names[s], np.nanmean(col), " ".join("%7.0f" % v for v in np.nanmean(col, axis=0))))
tot = np.nanmax(rel[:, :, 5], axis=1)
print("workgroup duration: mean %.0f  median %.0f  max %.0f cycles; %d workgroups -> %.1f per CU" % (tot.mean(), np.median(tot), tot.max(), len(t), len(t) / 256))
span = t[:, :, 5].max() - t[:, :, 0].min()
print("launch span (first start -> last end, one XCD clock domain assumed): %.0f cycles; sum of workgroup durations / 256 CUs = %.0f" % (span, tot.sum() / 256))
