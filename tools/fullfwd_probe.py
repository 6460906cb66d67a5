#!/usr/bin/env python3
"""The full forward from uint8 clips (ingest + glancer + policy + hot path, B = 64, T = 16, P = 96) for a rocprofv3 kernel trace:
`serial` = everything on one stream, `two` = GFV.offline_forward_pipelined (front half of batch i+1 and back half of batch i on the
model's two streams).  usage: python tools/fullfwd_probe.py serial|two [batches=10]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_extras import act_args, synth_model_state  # noqa: E402
from adafocus_amd.gfv_net import GFV  # noqa: E402
from adafocus_amd.transforms import ingest_uint8  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "serial"
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
b, t, p = 64, 16, 96
model = GFV(act_args(t, p, b)).eval()
model.load_state_dict(synth_model_state(model, 1007), strict=True)
model = model.to(dev)
clips = torch.randint(0, 256, (b, 224, 224, t * 3), dtype=torch.uint8, device=dev)


def run(n):
    for _ in range(n):
        if mode == "two":
            model.offline_forward_pipelined(clips, t)
        else:
            model.offline_forward_nhwc4(ingest_uint8(clips, t), b, t)
    if mode == "two":
        model.pipeline_flush()


with torch.no_grad():
    run(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(batches)
    torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / batches * 1e3
print("full forward, %s: %.3f ms per 64-clip batch = %.1f clips/s" % (mode, ms, b / ms * 1e3))
