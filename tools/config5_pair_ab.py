#!/usr/bin/env python3
"""Config 5's hot path (gather -> EfficientNet-B3 fp16 storage -> GRU classifier, B = 64, T = 16, P = 144) with the forward as one chunk / as a
pair of half chunks (option bit ADAF_EF_PLAN_PAIR_CHUNKS), from 1, 2 and 3 caller streams: clips/s, alternating rounds."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from adafocus_amd import _lib, synth  # noqa: E402
from adafocus_amd.gfv_net import GFV  # noqa: E402

dev = torch.device("cuda:0")
b, t, p = 64, 16, 144
frames = torch.from_numpy(synth.synth_frames(b, t, 224, seed=100)).to(dev).view(b * t, 3, 224, 224)
_, act_np = synth.synth_actions(b * t, 7, seed=5)
actions = torch.from_numpy(act_np).to(dev)
gvec = torch.randn((b, t, 1280), device=dev)
model = GFV(bench.act_args(t, p, b, local_arch="efficientnet-b3", local_dtype="f16")).eval()
model.load_state_dict(bench.synth_model_state(model, 1007), strict=True)
model = model.to(dev)
plan = int(_lib.get_option("effnet_plan"))
pools = {k: [torch.cuda.Stream(device=dev) for _ in range(k)] for k in (1, 2, 3)}


def rate(streams, steps=30):
    with torch.no_grad():
        for i in range(2 * len(streams)):
            with torch.cuda.stream(streams[i % len(streams)]):
                model.hot_path(frames, gvec, actions, b, t)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(steps):
            with torch.cuda.stream(streams[i % len(streams)]):
                model.hot_path(frames, gvec, actions, b, t)
        torch.cuda.synchronize()
    return steps * b / (time.perf_counter() - t1)


for rnd in range(3):
    row = []
    for k in (1, 2, 3):
        for on in (1, 0):
            with _lib.option("effnet_plan", plan | _lib.EF_PLAN_PAIR_CHUNKS if on else plan & ~_lib.EF_PLAN_PAIR_CHUNKS):
                row.append("%d stream(s) pair %s: %.0f" % (k, "on " if on else "off", rate(pools[k])))
    print("round %d: " % rnd + " | ".join(row), flush=True)
