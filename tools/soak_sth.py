#!/usr/bin/env python3
"""Soak of the Something-Something model (temporal shift everywhere: glancer strips with the shift in their pixel loads, lean shifted conv1 launches,
shifted next conv1 inside the fused stage-1 tail): the whole forward from uint8 clips repeated N times on the same input, on one stream and from two
streams at once; every output compared BIT FOR BIT with the first (a mismatch is a race).  usage: soak_sth.py [iters=200] [Tg=8] [Tf=8] [P=128]"""
import os
import sys
import threading

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_extras import sth_args, synth_model_state  # noqa: E402
from adafocus_amd.gfv_net_sth import GFV  # noqa: E402
from adafocus_amd.transforms import ingest_uint8  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tg = int(sys.argv[2]) if len(sys.argv) > 2 else 8
tf = int(sys.argv[3]) if len(sys.argv) > 3 else 8
p = int(sys.argv[4]) if len(sys.argv) > 4 else 128
dev = torch.device("cuda:0")
b = 64
a = sth_args(b, tg, p, tf=tf)
m = GFV(a).eval()
m.focuser.net.base_model = torch.nn.Sequential(*list(m.focuser.net.base_model.children())[:-1])     # STH/evaluate.py:83
m.load_state_dict(synth_model_state(m, 1007), strict=True)
m = m.to(dev)
gu = torch.randint(0, 256, (b, 224, 224, tg * 3), dtype=torch.uint8, device=dev)
fu = torch.randint(0, 256, (b, 224, 224, tf * 3), dtype=torch.uint8, device=dev)


def forward():
    g4 = ingest_uint8(gu, tg, m.input_mean, m.input_std)
    f4 = ingest_uint8(fu, tf, m.input_mean, m.input_std)
    fm4, glog = m.glance_nhwc4(g4, b)
    return m.action_stage2_nhwc4(f4, fm4, glog, 0, a, with_baseline=False)[0]


with torch.no_grad():
    ref = forward().clone()
    torch.cuda.synchronize()
    assert torch.isfinite(ref).all()
    bad = 0
    for i in range(iters):
        if not torch.equal(forward(), ref):
            bad += 1
    print("one stream: %d iterations, %d mismatches" % (iters, bad))
    bad2 = [0, 0]

    def worker(k):
        s = torch.cuda.Stream(device=dev)
        with torch.no_grad(), torch.cuda.stream(s):
            for i in range(iters // 2):
                out = forward()
                s.synchronize()
                if not torch.equal(out, ref):
                    bad2[k] += 1
    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    torch.cuda.synchronize()
    print("two streams at once: 2 x %d iterations, %d mismatches" % (iters // 2, sum(bad2)))
sys.exit(1 if bad or sum(bad2) else 0)
