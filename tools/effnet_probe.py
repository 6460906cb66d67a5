"""EfficientNet-B3 local CNN (BASELINE config 5) timing probe: N patches of P^2 through adaf_effnet in fp32 and fp16 storage.
usage: python tools/effnet_probe.py [N=1024] [P=144] [iters=20] [dtypes=f32,f16] [effnet_plan]
(effnet_plan: the library option of that name, e.g. 255 = everything but ADAF_EF_PLAN_PAIR_CHUNKS -- one chunk on one stream, so that a kernel
trace shows every kernel alone on the device)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import synth, workload  # noqa: E402
from adafocus_amd.efficientnet import EfficientNet  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
p = int(sys.argv[2]) if len(sys.argv) > 2 else 144
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dtypes = sys.argv[4].split(",") if len(sys.argv) > 4 else ["f32", "f16"]
if len(sys.argv) > 5:
    from adafocus_amd import _lib  # noqa: E402
    _lib.set_option("effnet_plan", int(sys.argv[5]))
dev = torch.device("cuda:0")
x4 = torch.randn((n, p, p, 4), device=dev)
x4[..., 3] = 0
res = {}
for dt in dtypes:
    m = EfficientNet.from_name("efficientnet-b3", num_classes=200, dtype=dt).eval()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1007).items()})
    m = m.to(dev)
    with torch.no_grad():
        for _ in range(3):
            m.features_nhwc4(x4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            m.features_nhwc4(x4)
        e1.record()
        torch.cuda.synchronize()
        host = (time.perf_counter() - t0) * 1e3 / iters
    ms = e0.elapsed_time(e1) / iters
    res[dt] = ms
    el = 2 if dt == "f16" else 4
    by, blk = workload.effnet_bytes_per_frame("efficientnet-b3", p, el), workload.effnet_block_bytes_per_frame("efficientnet-b3", p, el)
    print("effnet-b3 %s: %d x %d^2: %.3f ms (host %.3f ms)  %.0f patches/s  plan bytes %.2f MB/frame -> %.2f TB/s, block-level %.2f MB/frame -> %.2f TB/s"
          % (dt, n, p, ms, host, n / ms * 1e3, by / 1e6, by * n / ms / 1e9, blk / 1e6, blk * n / ms / 1e9))
if "f32" in res and "f16" in res:
    print("fp16 / fp32 storage speed-up: %.2fx" % (res["f32"] / res["f16"]))
