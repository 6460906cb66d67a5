#!/usr/bin/env python3
"""End-to-end rate of evaluate.validate on an in-memory synthetic set (host tensors -> pinned staging -> H2D -> full
forward -> metrics), fp32 clips as the reference's loader emits them vs the loader's uint8 clips."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import act_args, synth_model_state  # noqa: E402
from adafocus_amd import evaluate as E  # noqa: E402
from adafocus_amd.gfv_net import GFV  # noqa: E402

dev = torch.device("cuda:0")
t, b, n = 16, 64, int(os.environ.get("EVAL_N", "1024"))
args = act_args(t, 96, b)
args.gpu = 0
model = GFV(args).eval()
model.load_state_dict(synth_model_state(model, 1007), strict=True)
model = model.to(dev)
labels = torch.randint(0, 200, (n, 1))


class DS:
    def __init__(self, x):
        self.x = x

    def __len__(self):
        return n

    def __getitem__(self, i):
        return self.x[i % self.x.shape[0]], labels[i]


u8 = torch.randint(0, 256, (64, 224, 224, t * 3), dtype=torch.uint8)
f32 = torch.randn((64, t * 3, 224, 224))
for name, data in (("uint8 (H,W,T*3) clips", u8), ("fp32 (T*3,H,W) clips", f32)):
    E.validate(DS(data), model, torch.nn.CrossEntropyLoss(), args, quiet=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    E.validate(DS(data), model, torch.nn.CrossEntropyLoss(), args, quiet=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-24s %d clips in %.3f s = %.1f clips/s (%.1f ms per %d-clip batch)" % (name, n, dt, n / dt, 1e3 * dt * b / n, b), flush=True)
