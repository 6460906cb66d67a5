#!/usr/bin/env python3
"""Full forward from uint8 clips on ONE caller stream: the hot path (gather -> trunk -> GRU) as one 64-clip launch sequence vs two 32-clip halves on two
streams (what two caller streams give the bench's `value`, made inside one forward).  Alternating runs, outputs compared bit for bit."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench_extras import act_args, synth_model_state  # noqa: E402
from adafocus_amd.gfv_net import GFV  # noqa: E402
from adafocus_amd.transforms import ingest_uint8  # noqa: E402

dev = torch.device("cuda:0")
b, t, p = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 16, int(sys.argv[2]) if len(sys.argv) > 2 else 96
model = GFV(act_args(t, p, b)).eval()
model.load_state_dict(synth_model_state(model, 1007), strict=True)
model = model.to(dev)
clips = torch.randint(0, 256, (b, 224, 224, t * 3), dtype=torch.uint8, device=dev)
side = [torch.cuda.Stream(device=dev) for _ in range(2)]


def forward(parts):
    frames = ingest_uint8(clips, t)
    fmap, fvec = model.glancer.net.features_from_nhwc4(model.glancer_input(frames))
    table = model.focuser.action_table(dev)
    idx, actions = model.focuser.policy.policy_old.act_sequence_nhwc(fmap, b, t, table)
    g = fvec.view(b, t, -1)
    if parts == 1:
        return model.hot_path(frames, g, actions, b, t)[0]
    cur = torch.cuda.current_stream(dev)
    outs = []
    hb = b // parts
    for i in range(parts):
        s = side[i]
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(model.hot_path(frames[i * hb * t:(i + 1) * hb * t], g[i * hb:(i + 1) * hb], actions[i * hb * t:(i + 1) * hb * t], hb, t)[0])
    for s in side[:parts]:
        cur.wait_stream(s)
    return torch.cat(outs)


def ms(parts, n=10):
    for _ in range(3):
        forward(parts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        forward(parts)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    ref = forward(1).clone()
    two = forward(2).clone()
    print("two halves bit-identical to one pass:", torch.equal(ref, two), " max |d| %.3e" % (ref - two).abs().max().item())
    res = {1: [], 2: []}
    for _ in range(4):
        for parts in (1, 2):
            res[parts].append(ms(parts))
    for parts in (1, 2):
        print("hot path in %d part(s): %s  min %.3f ms = %.1f clips/s" % (parts, " ".join("%.3f" % v for v in res[parts]), min(res[parts]), b / min(res[parts]) * 1e3))
