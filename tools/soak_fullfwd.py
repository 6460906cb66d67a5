#!/usr/bin/env python3
"""Soak: the full forward from uint8 clips (ingest + glancer + policy + hot path) repeated N times on the same input; every output is compared
BIT FOR BIT with the first one (the path is deterministic: a mismatch is a race).  Both forms: serial on one stream, and the two-stream pipeline
(front half of batch i+1 beside the back half of batch i), whose concurrency moves the timing of every kernel.  Also reports the GRU scan's
time-out counter.  usage: python tools/soak_fullfwd.py [iters=300] [T=16] [P=96]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_extras import act_args, synth_model_state  # noqa: E402
from adafocus_amd import hip_ops  # noqa: E402
from adafocus_amd.gfv_net import GFV  # noqa: E402
from adafocus_amd.transforms import ingest_uint8  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
t = int(sys.argv[2]) if len(sys.argv) > 2 else 16
p = int(sys.argv[3]) if len(sys.argv) > 3 else 96
dev = torch.device("cuda:0")
b = 64
model = GFV(act_args(t, p, b)).eval()
model.load_state_dict(synth_model_state(model, 1007), strict=True)
model = model.to(dev)
clips = [torch.randint(0, 256, (b, 224, 224, t * 3), dtype=torch.uint8, device=dev) for _ in range(2)]
bad = 0
with torch.no_grad():
    ref = []
    for c in clips:
        out = model.offline_forward_nhwc4(ingest_uint8(c, t), b, t)
        ref.append([o.clone() for o in out])
    torch.cuda.synchronize()
    assert all(torch.isfinite(r[0]).all() for r in ref)
    for i in range(iters):
        k = i & 1
        out = model.offline_forward_nhwc4(ingest_uint8(clips[k], t), b, t)
        if not all(torch.equal(o, r) for o, r in zip(out, ref[k])):
            bad += 1
            print("serial: iteration %d differs (max |d logits| %.3e)" % (i, (out[0] - ref[k][0]).abs().max().item()), flush=True)
    print("serial: %d iterations, %d mismatches" % (iters, bad))
    pend, bad2 = [], 0
    for i in range(iters):
        k = i & 1
        logits, last, idx, done, _ = model.offline_forward_pipelined(clips[k], t)
        pend.append((k, logits, last, idx, done))
        if len(pend) > 2:
            kk, lg, ls, ix, dn = pend.pop(0)
            torch.cuda.current_stream(dev).wait_event(dn)
            if not (torch.equal(lg, ref[kk][0]) and torch.equal(ls, ref[kk][1]) and torch.equal(ix, ref[kk][3])):
                bad2 += 1
                print("pipelined: iteration %d differs (max |d logits| %.3e)" % (i - 2, (lg - ref[kk][0]).abs().max().item()), flush=True)
    model.pipeline_flush()
    for kk, lg, ls, ix, dn in pend:
        if not (torch.equal(lg, ref[kk][0]) and torch.equal(ls, ref[kk][1]) and torch.equal(ix, ref[kk][3])):
            bad2 += 1
    print("pipelined: %d iterations, %d mismatches" % (iters, bad2))
print("gru scan time-outs: %d" % hip_ops.gru_scan_timeouts(dev))
sys.exit(1 if bad or bad2 else 0)
