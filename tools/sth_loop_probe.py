#!/usr/bin/env python3
"""Where a Something-Something evaluation batch goes (evaluate.validate_sth from uint8 clips, T = 8 + 8, P = 128, B = 64): the GPU
work alone (clips resident, HIP events), the host staging alone, and the loop.  usage: python tools/sth_loop_probe.py [batches=8]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_extras as X  # noqa: E402
from adafocus_amd import evaluate as E  # noqa: E402
from adafocus_amd.gfv_net_sth import GFV  # noqa: E402
from adafocus_amd.transforms import ingest_uint8  # noqa: E402

batches = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
b, t, p = 64, 8, 128
a = X.sth_args(b, t, p)
m = GFV(a).eval()
m.focuser.net.base_model = torch.nn.Sequential(*list(m.focuser.net.base_model.children())[:-1])
m.load_state_dict(X.synth_model_state(m, 1007), strict=True)
m = m.to(dev)
g8 = torch.randint(0, 256, (b, 224, 224, t * 3), dtype=torch.uint8)
f8 = torch.randint(0, 256, (b, 224, 224, t * 3), dtype=torch.uint8)
gd, fd = g8.to(dev), f8.to(dev)


def gpu_batch(wb):
    g4 = ingest_uint8(gd, t, m.input_mean, m.input_std)
    f4 = ingest_uint8(fd, t, m.input_mean, m.input_std)
    fm4, glog = m.glance_nhwc4(g4, b)
    return m.action_stage2_nhwc4(f4, fm4, glog, 0, a, with_baseline=wb)


def ev_time(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


with torch.no_grad():
    print("GPU work per batch, clips resident: with baseline %.2f ms, without %.2f ms" % (ev_time(lambda: gpu_batch(True)), ev_time(lambda: gpu_batch(False))))
    g4 = ingest_uint8(gd, t, m.input_mean, m.input_std)
    print("  ingest x2 %.2f ms | glance %.2f ms" % (ev_time(lambda: (ingest_uint8(gd, t, m.input_mean, m.input_std), ingest_uint8(fd, t, m.input_mean, m.input_std))),
                                                   ev_time(lambda: m.glance_nhwc4(g4, b))))
    pin = torch.empty((2, b) + tuple(g8.shape[1:]), dtype=torch.uint8).pin_memory()
    t0 = time.perf_counter()
    for _ in range(5):
        pin[0].copy_(g8); pin[1].copy_(f8)
    print("host staging, one thread, 2 x %.0f MB: %.2f ms" % (g8.numel() / 1e6, (time.perf_counter() - t0) / 5 * 1e3))
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(8)
    t0 = time.perf_counter()
    for _ in range(5):
        list(pool.map(lambda i: (pin[0, i].copy_(g8[i]), pin[1, i].copy_(f8[i])), range(b)))
    print("host staging, 8 threads per clip: %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
    t0 = time.perf_counter()
    for _ in range(5):
        gd.copy_(pin[0], non_blocking=True); fd.copy_(pin[1], non_blocking=True)
        torch.cuda.synchronize()
    print("H2D 2 x %.0f MB from pinned: %.2f ms" % (g8.numel() / 1e6, (time.perf_counter() - t0) / 5 * 1e3))
    n = b * batches
    labels = torch.randint(0, 174, (n,))

    class DS:
        def __len__(self):
            return n

        def __getitem__(self, i):
            return g8[i % b], f8[(i + 5) % b], labels[i]
    crit = torch.nn.CrossEntropyLoss()
    for wb in (True, False):
        E.validate_sth(DS(), m, crit, a, quiet=True, with_baseline=wb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        E.validate_sth(DS(), m, crit, a, quiet=True, with_baseline=wb)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("validate_sth %s baseline: %.2f ms per batch = %.0f clips/s over %d batches" % ("with" if wb else "without", dt / batches * 1e3, n / dt, batches))
