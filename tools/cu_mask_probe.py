#!/usr/bin/env python3
"""Can the two halves of the full forward share the device?  The front half (uint8 ingest + MobileNetV2 glancer + policy: depthwise /
small-GEMM work, FMA-lane and HBM bound) of batch i+1 and the back half (gather + ResNet-50 trunk + GRU: MFMA bound) of batch i run on
two streams (GFV.offline_forward_pipelined).  This probe measures the serial forward, the two-stream forward on plain streams, and the
two-stream forward on CU-MASKED streams (hipExtStreamCreateWithCUMask: the front half confined to F CUs, the back half to the other
256 - F), for several splits, all on the same box.  usage: python tools/cu_mask_probe.py [batches=24] [--trace]  (--trace: fewer
iterations, for a rocprofv3 --kernel-trace run)"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_extras import act_args, synth_model_state  # noqa: E402
from adafocus_amd.gfv_net import GFV  # noqa: E402

trace = "--trace" in sys.argv
args = [v for v in sys.argv[1:] if not v.startswith("--")]
batches = int(args[0]) if args else (6 if trace else 24)
dev = torch.device("cuda:0")
b, t, p = 64, 16, 96
model = GFV(act_args(t, p, b)).eval()
model.load_state_dict(synth_model_state(model, 1007), strict=True)
model = model.to(dev)
clips = torch.randint(0, 256, (b, 224, 224, t * 3), dtype=torch.uint8, device=dev)
hip = C.CDLL("libamdhip64.so")
ncu = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(cus):
    """A stream whose kernels may only run on the listed CUs."""
    words = (ncu + 31) // 32
    mask = (C.c_uint32 * words)()
    for cu in cus:
        mask[cu // 32] |= 1 << (cu % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(words), mask)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed: %d" % rc)
    return torch.cuda.ExternalStream(s.value, device=dev)


def run_pipelined(n):
    outs = None
    for _ in range(n):
        outs = model.offline_forward_pipelined(clips, t)
    model.pipeline_flush()
    return outs


def timeit(fn, n):
    with torch.no_grad():
        fn(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(n)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def serial(n):
    from adafocus_amd.transforms import ingest_uint8
    for _ in range(n):
        model.offline_forward_nhwc4(ingest_uint8(clips, t), b, t)


def halves(n, which):
    from adafocus_amd.transforms import ingest_uint8
    fr = ingest_uint8(clips, t)
    fmap, fvec = model.glancer.net.features_from_nhwc4(fr)
    table = model.focuser.action_table(dev)
    idx, actions = model.focuser.policy.policy_old.act_sequence_nhwc(fmap, b, t, table)
    for _ in range(n):
        if which == "front":
            f2 = ingest_uint8(clips, t)
            fm, fv = model.glancer.net.features_from_nhwc4(f2)
            model.focuser.policy.policy_old.act_sequence_nhwc(fm, b, t, table)
        else:
            model.hot_path(fr, fvec.view(b, t, -1), actions, b, t)


rows = []
ms_serial = timeit(serial, batches)
rows.append(("serial, one stream", ms_serial))
rows.append(("front half alone (ingest + glancer + policy)", timeit(lambda n: halves(n, "front"), batches)))
rows.append(("back half alone (gather + trunk + GRU)", timeit(lambda n: halves(n, "back"), batches)))
model._pipe = None
rows.append(("two plain streams", timeit(run_pipelined, batches)))
# the glancer forks a library-owned (unmasked) side stream for its second frame chunk: one chunk at a time for the masked runs
model.glancer.net._engine.fusion = 1 | 4
model._pipe = None
rows.append(("two plain streams, glancer chunks one at a time", timeit(run_pipelined, batches)))
if not trace:
    for front_cus in (32, 64, 96, 128, 160):
        # interleaved assignment: CU c belongs to the front set when (c % 8) < front_cus / 32 -- every XCD / shader engine gives the same share
        k = front_cus // 32
        fset = [c for c in range(ncu) if (c % 8) < k]
        bset = [c for c in range(ncu) if (c % 8) >= k]
        try:
            model._pipe = (masked_stream(fset), masked_stream(bset))
            rows.append(("CU-masked streams, interleaved: front %d CUs / back %d" % (len(fset), len(bset)), timeit(run_pipelined, batches)))
            model._pipe = (masked_stream(list(range(front_cus))), masked_stream(list(range(front_cus, ncu))))
            rows.append(("CU-masked streams, contiguous: front %d CUs / back %d" % (front_cus, ncu - front_cus), timeit(run_pipelined, batches)))
        except Exception as exc:      # noqa: BLE001
            rows.append(("CU-masked streams front %d: %r" % (front_cus, exc), float("nan")))
    # how a half scales with its CU share (is either half close to linear in CUs, i.e. would partitioning even conserve work?)
    for cus in (64, 128, 192):
        s = masked_stream([c for c in range(ncu) if (c % 8) < cus // 32])
        for which in ("front", "back"):
            with torch.cuda.stream(s):
                ms = timeit(lambda n: halves(n, which), max(batches // 2, 4))
            rows.append(("%s half alone on %d CUs" % (which, cus), ms))
for name, ms in rows:
    print("%-70s %8.3f ms per 64-clip batch = %7.1f clips/s" % (name, ms, b / ms * 1e3))
