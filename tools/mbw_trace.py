#!/usr/bin/env python3
"""Phase timeline of the whole-block MBConv kernel (csrc/mbconv_whole.hip built with -DMBW_TRACE by tools/exp/build_mbw_trace.sh):
s_memtime stamps of every wave at the phase boundaries, averaged over the workgroups of the launch of block K-1.
usage: python tools/mbw_trace.py [block=14] [N=1024] [P=144]
MBW_CHUNKS=1 with the library of `MBW_EXTRA=-DMBW_TRACE_CHUNKS MBW_OUT=libadafocus_hip_mbwtrace_chunks.so tools/exp/build_mbw_trace.sh`:
the round-1 slots hold the hand-over of each B-fragment chunk of round 0 instead."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("ADAF_LIB", os.path.join(ROOT, "adafocus_amd", "csrc", "libadafocus_hip_mbwtrace%s.so" % ("_chunks" if os.environ.get("MBW_CHUNKS") else "")))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from adafocus_amd import _lib, synth  # noqa: E402
from adafocus_amd.efficientnet import EfficientNet  # noqa: E402

blk = int(sys.argv[1]) if len(sys.argv) > 1 else 14
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
lib = _lib.load_library()
raw = C.CDLL(_lib.LIB_PATH)
raw.adaf_mbw_set_trace.argtypes = [C.c_void_p, C.c_int, C.c_int]
buf = torch.zeros((n, 8, 16), dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(3):
        net.forward_blocks(x4, blk + 1)
    torch.cuda.synchronize()
    raw.adaf_mbw_set_trace(buf.data_ptr(), b["hid"], b["k"])
    net.forward_blocks(x4, blk + 1)
    torch.cuda.synchronize()
    raw.adaf_mbw_set_trace(None, 0, 0)
t = buf.cpu().numpy().astype(np.float64)
live = t[:, 0, 0] > 0
t = t[live]
print("workgroups traced: %d" % len(t))
t0 = t[:, :, 0].min(axis=1)[:, None, None]
rel = np.where(t > 0, t - t0, np.nan)
names = ["start", "X in LDS", "r0 products", "r0 swish+swap", "r0 depthwise", "r1 products", "r1 swish+swap", "r1 depthwise", "phase 1 done",
         "barrier (D complete)", "reduce FC done", "gate done", "barrier", "D gated", "barrier", "end"]
if os.environ.get("MBW_CHUNKS"):
    names[5:8] = ["(stamp-0 store done)", "first B loads issued", "first B loads arrived"]
print("%-22s %10s   per wave (mean over workgroups, cycles from the workgroup's first stamp)" % ("stamp", "mean"))
for s in range(16):
    col = rel[:, :, s]
    if np.all(np.isnan(col)):
        continue
    pw = np.nanmean(col, axis=0)
    print("%-22s %10.0f   %s" % (names[s], np.nanmean(col), " ".join("%7.0f" % v for v in pw)))
tot = np.nanmax(rel[:, :, 15], axis=1)
print("workgroup duration: mean %.0f  median %.0f  max %.0f cycles" % (tot.mean(), np.median(tot), tot.max()))
# when do the workgroups start (relative to the first of the launch): the rounds of residency
st = np.sort(t[:, 0, 0] - t[:, 0, 0].min())
print("workgroup start times, deciles: " + " ".join("%.0f" % st[int(q * (len(st) - 1))] for q in np.linspace(0, 1, 11)))
print("launch span (first start -> last end): %.0f cycles" % (t[:, :, 15].max() - t[:, 0, 0].min()))
