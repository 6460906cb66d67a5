#!/usr/bin/env python3
"""EfficientNet-B3 local CNN: the 1024 patches as ONE pass on one stream vs k interleaved sub-batches on k streams (do the
latency-bound launches of the 9^2 / 5^2 stages overlap?).  usage: python tools/effnet_streams_probe.py [dtype=f16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import synth  # noqa: E402
from adafocus_amd.efficientnet import EfficientNet  # noqa: E402

dt = sys.argv[1] if len(sys.argv) > 1 else "f16"
n, p = 1024, 144
dev = torch.device("cuda:0")
x4 = torch.randn((n, p, p, 4), device=dev)
x4[..., 3] = 0
m = EfficientNet.from_name("efficientnet-b3", num_classes=200, dtype=dt).eval()
shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1007).items()})
m = m.to(dev)
ref = None
with torch.no_grad():
    for k in (1, 2, 4):
        streams = [torch.cuda.Stream() for _ in range(k)]
        parts = [x4[i * (n // k):(i + 1) * (n // k)] for i in range(k)]
        outs = [None] * k

        def run():
            cur = torch.cuda.current_stream()
            for i, s in enumerate(streams):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    outs[i] = m.features_nhwc4(parts[i])
            for s in streams:
                cur.wait_stream(s)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        full = torch.cat(outs)
        if ref is None:
            ref = full
        print("%d stream(s) x %d patches: %.3f ms per 1024 patches; equal to the single pass: %s" % (k, n // k, e0.elapsed_time(e1) / 10, torch.equal(full, ref)))
