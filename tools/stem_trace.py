#!/usr/bin/env python3
"""Phase timeline of the EfficientNet stem kernel (csrc/effnet.hip ef_stem_kernel, trace build of tools/exp/build_mbw_trace.sh).
usage: python tools/stem_trace.py [N=1024] [P=144]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
p = int(sys.argv[2]) if len(sys.argv) > 2 else 144
dev = torch.device("cuda:0")
x4 = torch.randn((n, p, p, 4), device=dev)
x4[..., 3] = 0
m = EfficientNet.from_name("efficientnet-b3", num_classes=200, dtype="f16").eval()
shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1007).items()})
m = m.to(dev)
net = m.engine()
_lib.load_library()
raw = C.CDLL(_lib.LIB_PATH)
raw.adaf_ef_set_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
buf = torch.zeros((1 << 16, 4, 8), dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(3):
        net.forward_blocks(x4, 0)
    torch.cuda.synchronize()
    raw.adaf_ef_set_trace(buf.data_ptr(), 0, -1, 0)
    net.forward_blocks(x4, 0)
    torch.cuda.synchronize()
    raw.adaf_ef_set_trace(None, 0, 0, 0)
t = buf.cpu().numpy().astype(np.float64)
t = t[t[:, 0, 0] > 0]
print("workgroups traced: %d (%.1f per CU)" % (len(t), len(t) / 256))
t0 = t[:, :, 0].min(axis=1)[:, None, None]
rel = np.where(t > 0, t - t0, np.nan)
names = ["start", "filter bank in registers, first window requested", "tile 1: window in LDS (barrier)", "tile 1: products issued",
         "tile 1: BN + swish in the slab", "tile 1: stored", "tile 0 done", "end"]
for s in (0, 1, 6, 2, 3, 4, 5, 7):
    col = rel[:, :, s]
    print("%-52s %8.0f   %s" % (names[s], np.nanmean(col), " ".join("%7.0f" % v for v in np.nanmean(col, axis=0))))
tot = np.nanmax(rel[:, :, 7], axis=1)
print("workgroup duration: mean %.0f median %.0f max %.0f; sum / 256 CUs = %.0f cycles" % (tot.mean(), np.median(tot), tot.max(), tot.sum() / 256))
