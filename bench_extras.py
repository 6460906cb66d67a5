"""Extra rows of bench.py's JSON line (never `value`): the other BASELINE configurations, the resampling gather, the
evaluation loops end to end, small-batch latency.  Each function measures on `dev` with inputs resident in HBM (unless it
says otherwise) and returns a plain dict; bench.py wraps every call so that a failure here can never fail the bench.
"""
import time

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0


class Args:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def act_args(t, p, b, **over):
    a = Args(num_segments=t, num_classes=200, reward="random", dataset="actnet", input_size=224, batch_size=b,
             patch_size=p, with_glancer=True, feature_map_channels=1280, glance_size=224, action_dim=49,
             hidden_state_dim=1024, policy_conv=True, gpu=0, continuous=False, gamma=0.7, policy_lr=0.0003,
             random_patch=False, dropout=0.5, consensus="gru", hidden_dim=1024)
    a.__dict__.update(over)
    return a


def sth_args(b, t=8, p=128, video_div=1, tf=None):
    """Something-Something V1 configuration (STH/evaluate.py argparse defaults + the README's command line): TSM-MobileNetV2
    glancer over `t` frames, TSM-ResNet-50 focuser over `tf` (default: t) frames, continuous policy."""
    return Args(num_segments_glancer=t, num_segments_focuser=tf or t, num_classes=174, batch_size=b, patch_size=p, input_size=224,
                with_glancer=True, feature_map_channels=1280, video_div=video_div, glance_size=224, action_dim=49,
                hidden_state_dim=1024, policy_conv=True, gpu=0, ppo_continuous=True, gamma=0.7, policy_lr=0.0003,
                action_std=0.25, actorcritic_with_bn=True, modality="RGB", base_model="resnet50", partial_bn=False,
                pretrain="imagenet", is_shift=True, shift_div=8, shift_place="blockres", fc_lr5=False,
                temporal_pool=False, non_local=False, random_patch=False, dropout=0.5)


def synth_model_state(model, seed):
    from adafocus_amd import synth
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    return {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, seed).items()}


def _clock(fn, streams, steps, warm=None):
    warm = 2 * len(streams) if warm is None else warm
    for i in range(warm):
        with torch.cuda.stream(streams[i % len(streams)]):
            fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(streams[i % len(streams)]):
            fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def act_hot_path_row(dev, t, p, b, streams, steps):
    """ActivityNet hot path (gather -> ResNet-50 -> GRU classifier) at another (T, P): BASELINE config 2 (T=8, P=96) and
    config 3's per-GPU half (T=16, P=128)."""
    from adafocus_amd import synth, workload
    from adafocus_amd.gfv_net import GFV
    m = GFV(act_args(t, p, b)).eval()
    m.load_state_dict(synth_model_state(m, 1007), strict=True)
    m = m.to(dev)
    frames = torch.from_numpy(synth.synth_frames(b, t, 224, seed=1)).to(dev).view(b * t, 3, 224, 224)
    actions = torch.from_numpy(synth.synth_actions(b * t, 7, seed=2)[1]).to(dev)
    gvec = torch.randn((b, t, 1280), device=dev)
    with torch.no_grad():
        sec = _clock(lambda: m.hot_path(frames, gvec, actions, b, t), streams, steps)
    flop = workload.hot_path_flops_per_clip(t, p) * b
    return {"clips_per_s": round(b / sec, 1), "ms_per_step": round(sec * 1e3, 3), "steps": steps, "seconds": round(sec * steps, 3),
            "B": b, "T": t, "P": p, "streams": len(streams), "tflops": round(flop / sec / 1e12, 1),
            "frac_of_f32_mfma_peak": round(flop / sec / 1e12 / 157.3, 4)}


def sth_hot_path_row(dev, b, streams, steps, t=8, p=128, tg=None):
    """BASELINE config 4: Something-Something V1, TSM-ResNet-50 local CNN (temporal shift fused into every Bottleneck conv1's
    operand load), T = 8, P = 128: gather (one (y, x) per clip) -> TSM trunk -> FC + temporal mean + glancer logits
    (GFV.action_stage3 in eval mode with the action given: the hot path without its producers).  `t` = focuser frames per clip,
    `tg` = glancer frames (default t)."""
    from adafocus_amd import synth, workload
    from adafocus_amd.gfv_net_sth import GFV
    tg = tg or t
    a = sth_args(b, tg, p, tf=t)
    m = GFV(a).eval()
    m.focuser.net.base_model = torch.nn.Sequential(*list(m.focuser.net.base_model.children())[:-1])     # STH/evaluate.py:83
    m.load_state_dict(synth_model_state(m, 1007), strict=True)
    m = m.to(dev)
    fo = torch.from_numpy(synth.synth_frames(b, t, 224, seed=4)).view(b, t, 3, 224, 224).to(dev)
    fm = torch.randn((b, tg, 7, 7, 1280), device=dev).permute(0, 1, 4, 2, 3)     # glancer map, reference-layout view
    glog = torch.randn((b, tg, 174), device=dev)
    forced = torch.rand((b, 2), device=dev)
    with torch.no_grad():
        sec = _clock(lambda: m.action_stage3(fo, fm, glog, 0, a, prev_local_patch=None, forced_action=forced), streams, steps)
    flop = 2.0 * workload.resnet50_macs_per_patch(p) * b * t
    return {"clips_per_s": round(b / sec, 1), "ms_per_step": round(sec * 1e3, 3), "steps": steps, "seconds": round(sec * steps, 3),
            "B": b, "T": t, "P": p, "streams": len(streams), "tflops": round(flop / sec / 1e12, 1),
            "frac_of_f32_mfma_peak": round(flop / sec / 1e12 / 157.3, 4)}


def gather_resize_row(dev, frames, patch=96, iters=20):
    """Row N1: the resampling crop (adaf_crop_resize_f32: window of size S from (y, x) -> bilinear -> patch^2) at S != patch,
    priced against HBM.  Algorithmic bytes per patch = 3 S^2 4 (the window, read once) + 3 patch^2 4 (the patch, written
    once).  `frames` (N,3,H,W) fp32 resident on the device."""
    from adafocus_amd import hip_ops
    n = frames.shape[0]
    g = torch.Generator(device="cpu").manual_seed(11)
    actions = torch.rand((n, 2), generator=g).to(dev)
    out = {}

    def run(size):
        fn = lambda: hip_ops.crop_resize(frames, actions, patch, size=size, layout=hip_ops.LAYOUT_NHWC4)   # noqa: E731
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    for name, size in (("S128_to_P%d" % patch, 128), ("S192_to_P%d" % patch, 192)):
        ms = run(size)
        by = float(n) * (3 * size * size * 4 + 3 * patch * patch * 4)
        out[name] = {"ms": round(ms, 4), "achieved": round(by / ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(by / ms / 1e6 / HBM_PEAK_GBS, 4), "bytes_per_patch": int(by / n)}
    sizes = torch.tensor(np.random.Generator(np.random.PCG64(12)).choice([96, 128, 160, 192], size=n), dtype=torch.int32).to(dev)
    ms = run(sizes)
    by = float((3 * sizes.double() ** 2 * 4).sum().item()) + float(n) * 3 * patch * patch * 4
    out["mixed_S96_128_160_192_to_P%d" % patch] = {"ms": round(ms, 4), "achieved": round(by / ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                    "frac": round(by / ms / 1e6 / HBM_PEAK_GBS, 4), "bytes_per_patch": int(by / n)}
    out["note"] = ("adaf_crop_resize_f32, %d frames of %dx%d NCHW fp32 -> (N,%d,%d,4): per-action window size S, bilinear (align_corners="
                   "False) resample; bytes = window read + patch written (3 channels)" % (n, frames.shape[2], frames.shape[3], patch, patch))
    return out


def evaluate_loop_row(dev, model, args, b, t, batches=24):
    """Row f3 end to end: evaluate.validate on an in-memory synthetic set of uint8 clips (host tensors -> pinned staging ->
    H2D -> ingest + glancer + policy + hot path -> loss / accuracy / mAP on the host).  One untimed pass first (pinned
    buffers, scratch), then the timed pass.  A call carries ~50 ms that do not overlap with anything (the first batch's staging +
    copy, the last batch's drain, cal_map on the host): 8 batches measure 1.9-2.0 k clips/s, 24 batches 2.4 k, the steady state
    is ~25.5 ms per 64-clip batch (tools/eval_loop_probe.py with EVAL_N=2048)."""
    from adafocus_amd import evaluate as E
    n = b * batches
    labels = torch.randint(0, args.num_classes, (n, 1))
    u8 = torch.randint(0, 256, (b, 224, 224, t * 3), dtype=torch.uint8)

    class DS:
        def __len__(self):
            return n

        def __getitem__(self, i):
            return u8[i % b], labels[i]
    crit = torch.nn.CrossEntropyLoss()
    E.validate(DS(), model, crit, args, quiet=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    E.validate(DS(), model, crit, args, quiet=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 1), "unit": "clips/s", "clips": n, "seconds": round(dt, 3), "ms_per_batch": round(dt / batches * 1e3, 2),
            "note": "evaluate.validate (ACT/main_dist.py:307-422 stage-3 branch) from the loader's uint8 (H,W,T*3) clips in host memory: "
                    "pinned staging + H2D of batch i+1 under batch i's kernels, two-stream forward, metrics (accuracy, cal_map) on the host; "
                    "%d batches per call (a call carries ~50 ms of un-overlapped first-batch staging, drain and cal_map)" % batches}


def validate_sth_row(dev, b, t=8, p=128, batches=8, tf=None, fp32_clips=True):
    """Something-Something loop end to end (evaluate.validate_sth = STH/evaluate.py:165-226): two frame streams from host memory,
    glancer + continuous policy + gather + TSM-ResNet-50 (+ the reward-baseline branch, as the reference runs it) + FC / consensus,
    accuracy over the set.  From the loader's stacked uint8 clips (normalised on the GPU: 4x fewer bytes over PCIe; row f1) and from
    the reference's fp32 clips."""
    from adafocus_amd import evaluate as E
    from adafocus_amd.gfv_net_sth import GFV
    tf = tf or t
    a = sth_args(b, t, p, tf=tf)
    m = GFV(a).eval()
    m.focuser.net.base_model = torch.nn.Sequential(*list(m.focuser.net.base_model.children())[:-1])
    m.load_state_dict(synth_model_state(m, 1007), strict=True)
    m = m.to(dev)
    n = b * batches
    g = torch.Generator().manual_seed(3)
    labels = torch.randint(0, 174, (n,), generator=g)
    gl8 = torch.randint(0, 256, (16, 224, 224, t * 3), dtype=torch.uint8, generator=g)
    fo8 = torch.randint(0, 256, (16, 224, 224, tf * 3), dtype=torch.uint8, generator=g)
    gl = torch.randn((16, t * 3, 224, 224), generator=g) if fp32_clips else None
    fo = torch.randn((16, tf * 3, 224, 224), generator=g) if fp32_clips else None

    class DS:
        def __init__(self, x, y):
            self.x, self.y = x, y

        def __len__(self):
            return n

        def __getitem__(self, i):
            return self.x[i % 16], self.y[(i + 5) % 16], labels[i]
    crit = torch.nn.CrossEntropyLoss()
    out = {}
    for tag, ds in (("", DS(gl8, fo8)),) + ((("fp32_clips_", DS(gl, fo)),) if fp32_clips else ()):
        for name, base in (("with_baseline_branch", True), ("without_baseline_branch", False)):
            E.validate_sth(ds, m, crit, a, quiet=True, with_baseline=base)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            E.validate_sth(ds, m, crit, a, quiet=True, with_baseline=base)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out[tag + name] = {"value": round(n / dt, 1), "unit": "clips/s", "clips": n, "seconds": round(dt, 3), "ms_per_batch": round(dt / batches * 1e3, 2)}
    out["note"] = ("evaluate.validate_sth, T=%d + %d, P=%d, video_div=1, %d batches of %d clips per call; with_ / without_baseline_branch: from the "
                   "loader's stacked uint8 (H,W,T*3) clips in host memory (2 x %.0f MB per batch over PCIe, normalised on the GPU); fp32_clips_*: from the "
                   "reference's normalised fp32 (T*3,H,W) clips (2 x %.0f MB per batch); the baseline branch doubles the local-CNN work for a "
                   "logged-only reward" % (t, tf, p, batches, b, b * (t + tf) / 2 * 3 * 224 * 224 / 1e6, b * (t + tf) / 2 * 3 * 224 * 224 * 4 / 1e6))
    return out


def sth_shipped_row(dev, b, streams):
    """The reference's SHIPPED Something-Something evaluation configuration (STH/evaluate.sh:5-15, STH/conf/evaluate.yaml:29-30):
    num_segments_glancer = 8, num_segments_focuser = 12, patch_size = 144, video_div = 1 -- the only configuration with a published
    throughput (figure/sthsth.png, Table 3: AdaFocus-TSM 144^2, MobileNetV2 + ResNet-50, 8 + 12 frames: 143.8 videos/s at bs = 64 on
    an RTX 2080 Ti).  Rows: the hot path (gather -> 12-segment TSM-ResNet-50 -> FC + consensus), the whole model forward from uint8
    clips resident in HBM (ingest + TSM-MobileNetV2 glancer + continuous policy + hot path, without the reward baseline: what an
    inference throughput measures), and the evaluate.validate_sth loop from host memory."""
    from adafocus_amd import workload
    from adafocus_amd.gfv_net_sth import GFV
    from adafocus_amd.transforms import ingest_uint8
    tg, tf, p = 8, 12, 144
    out = {"T_glancer": tg, "T_focuser": tf, "P": p, "B": b,
           "published": {"value": 143.8, "unit": "videos/s", "hardware": "RTX 2080 Ti", "batch": 64, "source": "figure/sthsth.png",
                         "note": "context only: other hardware, the reference's PyTorch 1.8 / cuDNN per-step loop"}}
    out["hot_path"] = sth_hot_path_row(dev, b, streams, 30, t=tf, p=p, tg=tg)
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
        sec = _clock(forward, streams[:1], 10, warm=3)
    flop = 2.0 * (workload.resnet50_macs_per_patch(p) * tf + workload.mobilenetv2_macs_per_frame(224) * tg) * b
    out["full_forward_from_uint8"] = {"value": round(b / sec, 1), "unit": "videos/s", "ms_per_batch": round(sec * 1e3, 3), "streams": 1,
                                      "gflop_per_video": round(flop / b / 1e9, 2), "tflops": round(flop / sec / 1e12, 1),
                                      "vs_published_other_hardware": round(b / sec / 143.8, 1),
                                      "note": "uint8 clips resident in HBM -> ingest -> glance_nhwc4 -> action_stage2_nhwc4(with_baseline=False)"}
    del m, gu, fu
    torch.cuda.empty_cache()
    loop = validate_sth_row(dev, b, t=tg, p=p, batches=6, tf=tf, fp32_clips=False)
    out["validate_sth_loop"] = loop
    return out


def latency_rows(dev, cases=((1, 8), (2, 8), (2, 16)), p=96, iters=200):
    """Small-batch latency of the hot path (BASELINE config 1 is B = 2, T = 8; the reference's only published CPU figure is a
    bs = 1 latency): launches issued one by one from Python vs the same step replayed from a captured HIP graph
    (GFV.capture_hot_path).  The step is a chain of ~50 dependent launches whose K loops run in a handful of blocks."""
    from adafocus_amd import synth
    from adafocus_amd.gfv_net import GFV
    rows = {}
    for b, t in cases:
        m = GFV(act_args(t, p, b)).eval()
        m.load_state_dict(synth_model_state(m, 1007), strict=True)
        m = m.to(dev)
        fr = torch.from_numpy(synth.synth_frames(b, t, 224, seed=1)).to(dev).view(b * t, 3, 224, 224)
        act = torch.from_numpy(synth.synth_actions(b * t, 7, seed=2)[1]).to(dev)
        gv = torch.randn((b, t, 1280), device=dev)
        with torch.no_grad():
            for _ in range(5):
                lg = m.hot_path(fr, gv, act, b, t)[0]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                lg = m.hot_path(fr, gv, act, b, t)[0]
            torch.cuda.synchronize()
            eager = (time.perf_counter() - t0) / iters
            ref = lg.clone()
            # `exclusive`: the persistent GRU scan stays in the graph (the same kernels as the eager step; this process runs nothing beside it)
            g = m.capture_hot_path(b, t, exclusive=True, check_every=0)
            g(fr, gv, act)
            torch.cuda.synchronize()
            same = bool(torch.equal(g.logits, ref))
            t0 = time.perf_counter()
            for _ in range(iters):
                g.replay()
            torch.cuda.synchronize()
            graph = (time.perf_counter() - t0) / iters
            g.check()
            # the default capture: GRU in its launch-per-step form (no grid barrier: safe beside other graphs / eager hot paths)
            gs = m.capture_hot_path(b, t)
            gs(fr, gv, act)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                gs.replay()
            torch.cuda.synchronize()
            graph_safe = (time.perf_counter() - t0) / iters
            del gs
            # the same step with every conv on the engine's batched tiles (the round-2 plan): what the small-batch form buys
            trunk = m.focuser.net._sync()
            trunk.set_latency_rows(0)
            for _ in range(5):
                lg = m.hot_path(fr, gv, act, b, t)[0]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                lg = m.hot_path(fr, gv, act, b, t)[0]
            torch.cuda.synchronize()
            batched = (time.perf_counter() - t0) / iters
            same_bits = bool(torch.equal(lg, ref))
            trunk.set_latency_rows(-1)
        rows["B%d_T%d_P%d" % (b, t, p)] = {"eager_ms": round(eager * 1e3, 4), "graph_ms": round(graph * 1e3, 4),
                                            "graph_default_capture_ms": round(graph_safe * 1e3, 4),
                                            "clips_per_s": round(b / min(eager, graph), 1), "graph_bit_identical_to_eager": same,
                                            "batched_tiles_only_ms": round(batched * 1e3, 4),
                                            "speedup_over_batched_tiles": round(batched / min(eager, graph), 3),
                                            "bit_identical_to_batched_tiles": same_bits}
        del g, m
    rows["note"] = ("one hot-path step (gather + ResNet-50 + GRU classifier) per call, back to back, %d calls; eager = Python-issued "
                    "launches, graph = GFV.capture_hot_path(exclusive=True) replay (persistent GRU scan in the graph), graph_default_capture = the default capture (GRU launch per step: no grid barrier, may replay beside anything); convs with <= 1536 GEMM rows run on the small-batch form "
                    "(csrc/conv_lat.hip: v_mfma_f32_16x16x4_f32 chains in the engine's k order), batched_tiles_only_ms = the same step "
                    "with that switched off (adaf_resnet50_set_latency_rows(net, 0))" % iters)
    return rows


def _events_ms(fn, iters=5, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, out


def split_bf16_row(dev, model, frames, gvec, actions, b, t, p, streams, steps, step_fn):
    """`also.split_bf16`: the SAME hot path with the local CNN's convolutions on the bf16 matrix pipe -- every fp32 operand split
    exactly into three bf16 parts, the six products of magnitude >= 2^-24 |xy| accumulated in fp32 (ADAF_MATH_F32_SPLIT_BF16;
    include/adafocus.h).  NOT the reported configuration (`value` stays on the exact fp32 pipe).  Roofline: the work priced is what the
    pipe actually executes -- 6 x the algorithmic FLOP -- against the dense bf16 MFMA peak (2.5 PFLOP/s); pricing the algorithmic FLOP
    against the fp32 peak would read > 1."""
    from adafocus_amd import workload
    from adafocus_amd.utils import get_patch_nhwc4
    net = model.focuser.net
    with torch.no_grad():
        ref_logits = model.hot_path(frames, gvec, actions, b, t)[0].clone()
        x4 = get_patch_nhwc4(frames, actions, p)
        feat32 = net.features_nhwc4(x4).clone()
        net.set_math("split_bf16")
        try:
            alt_logits = model.hot_path(frames, gvec, actions, b, t)[0].clone()
            feat_sp = net.features_nhwc4(x4).clone()
            for i in range(2 * len(streams)):
                step_fn(i)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(steps):
                step_fn(i)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / steps
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            with torch.cuda.stream(streams[0]):
                for i in range(steps):
                    model.hot_path(frames, gvec, actions, b, t)
            torch.cuda.synchronize()
            dts = (time.perf_counter() - t1) / steps
            trunk = net._sync()
            trunk.forward(x4)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                trunk.forward(x4)
            e1.record()
            torch.cuda.synchronize()
            trunk_ms = e0.elapsed_time(e1) / 3
        finally:
            net.set_math("f32")
    # error against fp64 of both pipes on a sample of patches (the oracle-free statement: which pipe is closer to exact arithmetic)
    alg_flop = 2.0 * workload.resnet50_macs_per_patch(p) * b * t
    stem_flop = 2.0 * (p // 2) ** 2 * 64 * 147 * b * t            # the stem keeps the fp32 pipe
    pipe_flop = 6.0 * (alg_flop - stem_flop)
    ach = pipe_flop / (trunk_ms * 1e-3) / 1e12
    return {"clips_per_s": round(b / dt, 1), "ms_per_step": round(dt * 1e3, 3), "streams": len(streams), "steps": steps,
            "serial": {"clips_per_s": round(b / dts, 1), "ms_per_step": round(dts * 1e3, 3)},
            "max_abs_logit_diff_vs_f32": float((alt_logits - ref_logits).abs().max()),
            "max_abs_feature_diff_vs_f32": float((feat_sp - feat32).abs().max()), "feature_scale": float(feat32.abs().max()),
            "roofline": {"bound": "mfma", "pipe": "bf16 (v_mfma_f32_32x32x16_bf16), fp32 accumulate", "achieved": round(ach, 1), "peak": 2500.0,
                         "unit": "TFLOP/s", "frac": round(ach / 2500.0, 4), "trunk_ms": round(trunk_ms, 3),
                         "work_is": "6 bf16 products per algorithmic multiply-add of the non-stem convs (the stem stays on the fp32 pipe)",
                         "algorithmic_tflops": round(alg_flop / (trunk_ms * 1e-3) / 1e12, 1),
                         "note": "the algorithmic rate exceeds what the fp32 pipe can do at all (157.3 TF) when it reads > 157"},
            "note": "ADAF_MATH_F32_SPLIT_BF16 (opt-in; ResNet.set_math('split_bf16')): fp32 in, fp32 out, fp32 accumulate; every GPU parity module "
                    "also runs in this mode (tests/conftest.py trunk_math)"}
