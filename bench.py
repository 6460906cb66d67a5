#!/usr/bin/env python3
"""Headline benchmark: clips/s of the AdaFocus offline-inference hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  N > 1 from a bare shell: bench.py re-executes itself under `python -m torch.distributed.run --nnodes=1
  --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` (one rank per GPU, RCCL), the counterpart of the
  reference's own `mp.spawn` + `init_process_group` (ACT/main_dist.py:59,79-80); launched under torchrun already
  (WORLD_SIZE set), it just joins.  `--dry-run` exercises the same launch / barrier / max-over-ranks / gather / JSON
  path on CPU tensors over gloo with a stand-in step (tests/test_bench_launch.py) -- its `value` is meaningless.

A "step" = one pass of the hot path over one batch of B clips per GPU, inputs resident in HBM:
batched patch gather of B*T windows from (B*T,3,224,224) frames -> ResNet-50 local CNN over the
B*T patches -> concat with the glancer's 1280-d vectors -> GRU + FC classifier (+ for N > 1 one RCCL
all-gather of the (B,C) logits).  Workload = BASELINE.json's metric configuration: T=16 frames,
96x96 patches, ResNet-50 local CNN, B=64 clips per GPU (weak scaling: clips shard across ranks).
Glancer features and policy actions are producers upstream of the path and are generated once
before the timed region (the actions are a forced uniform-random 7x7 grid sequence so the gather
addresses scatter).

Prints ONE JSON line (rank 0) with the contract fields plus:
  roofline     -- dominant kernel (implicit-GEMM conv on fp32 MFMA): algorithmic FLOP / HIP-event time
  cpu_baseline -- the oracle (torch-CPU restatement of the reference) timed on this box's host cores
"""
import argparse
import json
import os
import socket
import statistics
import sys
import time

# The GPU boxes' host driver only supports dmabuf IPC: without this RCCL (and any cross-process device-memory sharing) fails
# with `hipIpcGetMemHandle: invalid argument`.  The image exports it already; `setdefault` keeps a bare `python bench.py
# --gpus N` working from an environment that was rebuilt without it.  It has to be in place before HIP initialises.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: dense f32 MFMA peak
HBM_PEAK_GBS = 8000.0


import bench_extras as X  # noqa: E402
from bench_extras import Args, act_args, synth_model_state  # noqa: E402,F401


def config5_row(dev, b, streams, frames, steps=30):
    """BASELINE config 5 as named: EfficientNet-B3 local CNN, T = 16, P = 144, fp16 storage -- the hot path (gather ->
    adafocus_amd.efficientnet on csrc/effnet.hip -> GRU classifier), with fp32 storage beside it, and the local CNN alone
    priced against HBM.  No reference implementation exists (EfficientNet is a dead import there):
    parity unpinned (checker = oracle/ref_effnet.py, the published algorithm of efficientnet_pytorch); never the headline."""
    from adafocus_amd import synth, workload
    from adafocus_amd.gfv_net import GFV
    from adafocus_amd.utils import get_patch_nhwc4
    t, p = 16, 144
    out = {}
    _, act_np = synth.synth_actions(b * t, 7, seed=5)
    actions = torch.from_numpy(act_np).to(dev)
    gvec = torch.randn((b, t, 1280), device=dev)
    ref = None
    for dtype in ("f32", "f16"):
        model = GFV(act_args(t, p, b, local_arch="efficientnet-b3", local_dtype=dtype)).eval()
        model.load_state_dict(synth_model_state(model, 1007), strict=True)
        model = model.to(dev)
        with torch.no_grad():
            lg = model.hot_path(frames, gvec, actions, b, t)[0].clone()
            for i in range(2 * len(streams)):            # every stream allocates its scratch before the clock starts
                with torch.cuda.stream(streams[i % len(streams)]):
                    model.hot_path(frames, gvec, actions, b, t)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(steps):
                with torch.cuda.stream(streams[i % len(streams)]):
                    model.hot_path(frames, gvec, actions, b, t)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            # the local CNN alone, HIP events on its stream, against the bytes of the launch plan that runs
            x4 = get_patch_nhwc4(frames, actions, p)
            net = model.focuser.net
            for _ in range(2):
                net.features_nhwc4(x4)
            cnn_ms = float("inf")
            for _ in range(3):                 # best of three groups of five (a single group has read 15 % high on a busy box)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    net.features_nhwc4(x4)
                e1.record()
                torch.cuda.synchronize()
                cnn_ms = min(cnn_ms, e0.elapsed_time(e1) / 5)
            # the same forward as ONE chunk on one stream (the default cuts a batch of >= 512 patches into two half chunks that travel side by
            # side on two streams, ADAF_EF_PLAN_PAIR_CHUNKS; bit-identical): what the pairing buys, measured in this run
            from adafocus_amd import _lib as _L
            one_ms = float("inf")
            with _L.option("effnet_plan", int(_L.get_option("effnet_plan")) & ~_L.EF_PLAN_PAIR_CHUNKS):
                for _ in range(2):
                    net.features_nhwc4(x4)
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        net.features_nhwc4(x4)
                    e1.record()
                    torch.cuda.synchronize()
                    one_ms = min(one_ms, e0.elapsed_time(e1) / 5)
        elem = 2 if dtype == "f16" else 4
        by = float(workload.effnet_block_bytes_per_frame("efficientnet-b3", p, elem)) * b * t      # block-level algorithmic bytes
        plan = float(workload.effnet_bytes_per_frame("efficientnet-b3", p, elem)) * b * t            # in + out of every launch that runs
        struct_by = float(workload.effnet_structural_bytes_per_frame("efficientnet-b3", p, elem)) * b * t
        out[dtype + "_storage"] = {
            "clips_per_s": round(steps * b / dt, 1), "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
            "local_cnn": {"bound": "hbm", "ms": round(cnn_ms, 3), "achieved": round(by / cnn_ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(by / cnn_ms / 1e6 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_patch": int(by / (b * t)),
                          "structural_bytes_per_patch": int(struct_by / (b * t)), "structural_frac": round(struct_by / cnn_ms / 1e6 / HBM_PEAK_GBS, 4),
                          "structural_bytes_are": "algorithmic_bytes_per_patch (the floor of a network WITHOUT squeeze-and-excite) + one write and one read of the "
                                                  "depthwise map of every block whose map of one image exceeds a CU's 160 KB of LDS (the gate needs the whole "
                                                  "map's average before the project conv can start): the floor an SE network can reach "
                                                  "(workload.effnet_structural_bytes_per_frame)",
                          "plan_bytes_per_patch": int(plan / (b * t)), "plan_gbs": round(plan / cnn_ms / 1e6, 1),
                          "plan_frac": round(plan / cnn_ms / 1e6 / HBM_PEAK_GBS, 4),
                          "traffic_bytes_per_patch": load_effnet_traffic(dtype, b * t, p),
                          "ms_one_chunk": round(one_ms, 3), "chunk_pairs": bool(int(_L.get_option("effnet_plan")) & _L.EF_PLAN_PAIR_CHUNKS),
                          "whole_block_launches": int(net.engine().whole_blocks(p)),
                          "fused_expand_launches": int(net.engine().fused_expand_blocks(p)),
                          "tflops": round(2.0 * workload.effnet_macs_per_frame("efficientnet-b3", p) * b * t / cnn_ms / 1e9, 1)}}
        if ref is None:
            ref = lg
        else:
            out[dtype + "_storage"]["rel_rms_logit_diff_vs_f32_storage"] = float(((lg - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item())
        del model
    out["fp16_over_fp32_storage_speedup"] = {"hot_path": round(out["f16_storage"]["clips_per_s"] / out["f32_storage"]["clips_per_s"], 3),
                                             "local_cnn": round(out["f32_storage"]["local_cnn"]["ms"] / out["f16_storage"]["local_cnn"]["ms"], 3)}
    out["gmac_per_patch"] = round(workload.effnet_macs_per_frame("efficientnet-b3", p) / 1e9, 4)
    out["note"] = ("gather (P=144) + EfficientNet-B3 (MBConv + squeeze-excite + swish, 3x3 / 5x5 depthwise, 26 blocks, 1536-d features) over "
                   "B*T = %d patches + GRU classifier, B=%d, T=16; fp16 = activations and 1x1 filters stored as fp16, fp32 accumulate; "
                   "local_cnn = the network alone (HIP events): achieved / frac are priced on the BLOCK-LEVEL algorithmic bytes (every tensor that crosses a block "
                   "boundary written once and read once + identity rows + patch in + feature out: workload.effnet_block_bytes_per_frame), plan_* on the activation "
                   "in + out of every launch of the plan that runs (workload.effnet_bytes_per_frame; whole_block_launches of the 26 MBConv blocks are one launch "
                   "each, csrc/mbconv_whole.hip; fused_expand_launches more compute their expand conv inside the depthwise launch, csrc/effnet.hip XN > 0), traffic_bytes_per_patch = 2 x FETCH_SIZE + WRITE_SIZE from the committed rocprofv3 passes (null if none); "
                   "parity unpinned (no reference implementation of this config)" % (b * t, b))
    return out


def init_failure_line(a, world, rank, backend, exc, launched=True):
    """The first unattended multi-GPU run must leave something diagnosable: rank 0 (or, failing that, whichever rank gets here) prints ONE
    JSON line with the bench contract's keys, `error`, `rccl_ranks: 0`, the exception text and the HSA_* / NCCL_* / RCCL_* / rendezvous
    environment; the other ranks say the same on stderr.  The process then exits non-zero (the counterpart of the reference's
    mp.spawn + init_process_group, ACT/main_dist.py:59,79-80, which dies with a traceback per rank)."""
    env = {k: v for k, v in sorted(os.environ.items())
           if k.startswith(("HSA_", "NCCL_", "RCCL_", "HIP_", "ROCR_", "MASTER_", "GLOO_", "TORCH_NCCL", "TORCH_DIST")) or k in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    line = {"metric": "clips/sec (T=%d, patch=%d^2, ResNet-50 local)" % (a.frames, a.patch), "value": None, "unit": "clips/s", "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": None,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "error": "distributed init failed on rank %d of %d (backend %s)" % (rank, world, backend), "rccl_ranks": 0,
            "exception": ("%s: %s" % (type(exc).__name__, exc))[:2000],
            "visible_gpus": (torch.cuda.device_count() if torch.cuda.is_available() else 0), "env": env}
    line["env"]["HIP_VISIBLE_DEVICES"] = os.environ.get("HIP_VISIBLE_DEVICES")
    line["env"]["ROCR_VISIBLE_DEVICES"] = os.environ.get("ROCR_VISIBLE_DEVICES")
    text = json.dumps(line)
    if not launched:          # the parent, before any rank exists
        print(text, flush=True)
        return
    if rank == 0:
        print(text, flush=True)
        time.sleep(1.0)       # the launcher tears the group down at the first non-zero exit: let the other ranks' stderr lines out first
    else:
        print("bench.py rank %d: %s" % (rank, text), file=sys.stderr, flush=True)
        time.sleep(3.0)       # ... and never before rank 0 has had the time to print THE line


def load_effnet_traffic(dtype, patches, p):
    """HBM bytes per patch of the EfficientNet forward from the committed rocprofv3 PMC passes (profiles/r<N>_effnet_traffic.json, written by
    tools/publish_profiles.py from separate FETCH_SIZE / WRITE_SIZE runs of tools/effnet_probe.py); null when not collected for this case."""
    for tag in ("r5", "r4"):
        path = os.path.join(ROOT, "profiles", "%s_effnet_traffic.json" % tag)
        if os.path.exists(path):
            try:
                j = json.load(open(path))
                e = j.get(dtype)
                if e and e.get("patches") == patches and e.get("P") == p:
                    return int(e["bytes_per_patch"])
            except Exception:
                pass
    return None


def load_glancer_traffic(frames):
    """HBM bytes per frame of the glancer from the committed rocprofv3 PMC passes (profiles/r<N>_glancer_traffic.json: separate FETCH_SIZE /
    WRITE_SIZE runs of tools/glancer_probe.py, tools/publish_profiles.py); null when not collected for this frame count."""
    for tag in ("r6", "r5", "r4"):
        path = os.path.join(ROOT, "profiles", "%s_glancer_traffic.json" % tag)
        if os.path.exists(path):
            try:
                j = json.load(open(path))
                if j.get("frames") == frames:
                    return int(j["bytes_per_frame"])
            except Exception:
                pass
    return None


def load_traffic(t, p, b):
    """HBM bytes per conv launch from the committed rocprofv3 PMC passes (profiles/r<N>_traffic.json, tools/profile_bench.sh +
    tools/publish_profiles.py; newest round first); only reported when it was collected on this very workload."""
    for tag in ("r6", "r5", "r4", "r3", "r2"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", "%s_traffic.json" % tag)))
            w = d["workload"]
            if (w["frames"], w["patch"], w["batch"]) == (t, p, b):
                return d["hbm_bytes_per_launch"]
        except Exception:
            pass
    return None


def host_cpu():
    """(model string, physical cores, logical CPUs) of the host, from /proc/cpuinfo."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core))
                phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    return model, (len(cores) or None), (os.cpu_count() or 1)


def numa_cpu_sets():
    """[(node, sorted physical-core representatives, all logical cpus)] from sysfs; one entry for the whole machine if there
    is no NUMA information.  A physical core is represented by the lowest-numbered of its hardware threads."""
    nodes = []
    try:
        for d in sorted(os.listdir("/sys/devices/system/node")):
            if not d.startswith("node") or not d[4:].isdigit():
                continue
            cpus = set()
            for part in open("/sys/devices/system/node/%s/cpulist" % d).read().strip().split(","):
                if part:
                    lo, _, hi = part.partition("-")
                    cpus.update(range(int(lo), int(hi or lo) + 1))
            cpus &= os.sched_getaffinity(0)
            reps = set()
            for c in cpus:
                try:
                    sib = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip()
                    reps.add(int(sib.replace("-", ",").split(",")[0]))
                except OSError:
                    reps.add(c)
            if cpus:
                nodes.append((int(d[4:]), sorted(reps & cpus) or sorted(cpus), sorted(cpus)))
    except OSError:
        pass
    if not nodes:
        allc = sorted(os.sched_getaffinity(0))
        nodes = [(0, allc, allc)]
    return nodes


def cpu_worker(spec):
    """`bench.py --cpu-worker T,P,B,threads,seconds,cpu,cpu,...`: one process of the "all cores" CPU row -- pins itself to the
    given CPUs, runs the oracle's batched hot path for ~seconds after two warm-ups, prints {"clips_per_s": ...}."""
    vals = [int(v) for v in spec.split(",")]
    t, p, b, threads, seconds, cpus = vals[0], vals[1], vals[2], vals[3], vals[4], vals[5:]
    if cpus:
        os.sched_setaffinity(0, set(cpus))
    torch.set_num_threads(threads)
    from adafocus_amd import synth
    from oracle import ref_model as O
    from tests.helpers import manifest
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(manifest()["ACT"], 1007).items()}
    frames = torch.from_numpy(synth.synth_frames(b, t, 224, seed=1)).view(b * t, 3, 224, 224)
    act = torch.from_numpy(synth.synth_actions(b * t, 7, seed=2)[1])
    gvec = torch.randn(b, t, 1280)
    with torch.no_grad():
        for _ in range(2):
            O.act_hot_path(sd, frames, gvec, act, p)
        n, t0 = 0, time.perf_counter()
        while n < 3 or time.perf_counter() - t0 < seconds:
            O.act_hot_path(sd, frames, gvec, act, p)
            n += 1
        dt = time.perf_counter() - t0
    print(json.dumps({"clips_per_s": n * b / dt, "runs": n}), flush=True)


def cpu_all_cores_row(t, p, threads, seconds=6):
    """The whole host: one process per group of `threads` physical cores (each pinned inside ONE NUMA node), all running the
    oracle's batched hot path at B = 2 at the same time; the row is the SUM of their rates."""
    import subprocess
    groups = []
    for _, phys, _ in numa_cpu_sets():
        for i in range(0, len(phys) - threads + 1, threads):
            groups.append(phys[i:i + threads])
    groups = groups[:16]
    if not groups:
        return None
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker",
                               ",".join(str(v) for v in [t, p, 2, threads, seconds] + g)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
             for g in groups]
    rates = []
    for pr in procs:
        out, _ = pr.communicate(timeout=300)
        for ln in out.splitlines():
            if ln.startswith("{"):
                rates.append(json.loads(ln)["clips_per_s"])
    if not rates:
        return None
    return {"clips_per_s": round(sum(rates), 2), "processes": len(rates), "threads_per_process": threads, "cores": len(rates) * threads,
            "B": 2, "T": t, "P": p, "structure": "batched, %d processes x %d threads pinned to disjoint physical cores, concurrent" % (len(rates), threads),
            "per_process_clips_per_s": [round(r, 2) for r in rates]}


def cpu_baseline_leg(sd, t, p, threads):
    """Oracle (port of the reference's PyTorch-CPU path, pinned to it by tests/golden) on the host, SURVEY.md §8(d)
    protocol: fp32, `torch.set_num_threads(n)`, 2 warm-ups + the MEDIAN of 7 runs with `time.perf_counter`, at B = 2
    (BASELINE config 1's batch) and B = 8, batched structure and the reference's per-step loop structure, for the
    metric's (T, P) and for the other ResNet-50 configurations.  oneDNN convs on the GPU boxes' hosts peak at a modest
    thread count (measured: 8 threads beat 128 by 5x), so n is chosen first by a short sweep at B = 8 and stated."""
    from adafocus_amd import synth
    from oracle import ref_model as O
    model, phys, ncpu = host_cpu()

    def inputs(nc, tt):
        frames = torch.from_numpy(synth.synth_frames(nc, tt, 224, seed=1)).view(nc * tt, 3, 224, 224)
        _, actions = synth.synth_actions(nc * tt, 7, seed=2)
        return frames, torch.from_numpy(actions), torch.randn(nc, tt, 1280)

    def batched(nc, tt, pp):
        frames, act, gvec = inputs(nc, tt)
        return lambda: O.act_hot_path(sd, frames, gvec, act, pp)

    def per_step(nc, tt, pp):
        # the reference's own loop (ACT/models/gfv_net.py:110-121): one crop + one local-CNN call per time step, then the GRU
        frames, act, gvec = inputs(nc, tt)
        fr, ac = frames.view(nc, tt, 3, 224, 224), act.view(nc, tt, 2)

        def run():
            feats = []
            for ti in range(tt):
                patch = O.get_patch(fr[:, ti].contiguous(), ac[:, ti].contiguous(), pp)
                feats.append(O.resnet50_trunk(sd, "focuser.net.", patch).view(nc, 1, -1))
            return O.recurrent_classifier(sd, "classifier.", torch.cat([gvec, torch.cat(feats, dim=1)], dim=2))
        return run

    def median_rate(fn, nc, warm=2, runs=7):
        for _ in range(warm):
            fn()
        times = []
        for _ in range(runs):
            t0 = time.perf_counter()
            fn()
            times.append(time.perf_counter() - t0)
        med = statistics.median(times)
        return round(nc / med, 3), round(med, 4)

    rows = []
    # the single-process rows run inside ONE NUMA node (its physical cores): a thread team that straddles sockets is what made
    # B = 8 slower than B = 2 on the 2-socket hosts
    old_aff = os.sched_getaffinity(0)
    node0 = numa_cpu_sets()[0]
    try:
        os.sched_setaffinity(0, set(node0[1]))
    except OSError:
        pass
    ncpu = min(ncpu, len(node0[1]))
    with torch.no_grad():
        sweep = {}
        if threads > 0:
            best_thr = threads
        else:
            fn = batched(8, t, p)
            for th in sorted({min(c, ncpu) for c in (8, 16, 32, (phys or ncpu))}):
                torch.set_num_threads(th)
                sweep[th] = median_rate(fn, 8, warm=1, runs=3)[0]
            best_thr = max(sweep, key=sweep.get)
        torch.set_num_threads(best_thr)
        for tt, pp in ((t, p),) + tuple(c for c in ((8, 96), (16, 128)) if c != (t, p)):
            for nc in (2, 8):
                rate, med = median_rate(batched(nc, tt, pp), nc)
                rows.append({"T": tt, "P": pp, "B": nc, "structure": "batched", "clips_per_s": rate, "median_s": med})
                if (tt, pp) == (t, p):
                    rate, med = median_rate(per_step(nc, tt, pp), nc)
                    rows.append({"T": tt, "P": pp, "B": nc, "structure": "reference loop (per time step)", "clips_per_s": rate,
                                 "median_s": med})
    try:
        os.sched_setaffinity(0, old_aff)
    except OSError:
        pass
    allrow = None
    try:
        allrow = cpu_all_cores_row(t, p, min(16, best_thr))
    except Exception as exc:
        allrow = {"error": repr(exc)[:200]}
    head = [r for r in rows if (r["T"], r["P"]) == (t, p)]
    best = max(head, key=lambda r: r["clips_per_s"])
    return {"value": best["clips_per_s"], "unit": "clips/s", "cores": best_thr, "kind": "port",
            "cpu_model": model, "physical_cores": phys, "logical_cpus": ncpu, "threads": best_thr,
            "thread_sweep_B8_clips_per_s": sweep or None, "value_is": "B=%d, %s" % (best["B"], best["structure"]),
            "pinned_to_numa_node": node0[0], "all_cores": allrow,
            "all_cores_note": "N processes x `threads` pinned to disjoint physical cores, run at the same time and summed -- reported as measured "
                              "(shared hosts: a container CPU quota and / or the other tenants' memory traffic bound it)",
            "rows": rows,
            "sample": "oracle.act_hot_path (crop -> ResNet-50 -> GRU; torch-CPU fp32 restatement of the reference, pinned by "
                      "tests/golden), T=%d P=%d, B=2 and B=8 clips per call, %d threads, 2 warm-ups + median of 7 runs; `value` = the "
                      "best of those rows; other configs in `rows`" % (t, p, best_thr)}


# glibc hands every tensor above 128 KB straight to mmap / munmap: a forward of the oracle then page-faults (and zeroes) hundreds of MB per
# call, under the mm lock, with a thread team waiting -- on the 256-thread hosts that made B = 8 (4x larger maps) 15x slower per call than
# B = 2.  With the heap kept (no mmap, no trim) the per-clip rate is flat in B; measured in the build container (8 cores,
# T = 16, P = 96): B = 2 11.8 -> 14.4 clips/s, B = 8 8.1 -> 14.3.  The variables are read at process start, hence the subprocess.
MALLOC_ENV = {"MALLOC_MMAP_THRESHOLD_": str(1 << 32), "MALLOC_TRIM_THRESHOLD_": str(1 << 33), "MALLOC_TOP_PAD_": str(1 << 30)}


def cpu_baseline(t, p, threads):
    """The CPU baseline leg in its own process (so that MALLOC_ENV applies): `bench.py --cpu-baseline-leg T,P,threads` prints the
    dictionary cpu_baseline_leg() returns as one JSON line."""
    import subprocess
    env = dict(os.environ)
    env.update(MALLOC_ENV)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-leg", "%d,%d,%d" % (t, p, threads)],
                         capture_output=True, text=True, timeout=900, env=env)
    for ln in out.stdout.splitlines():
        if ln.startswith("{"):
            res = json.loads(ln)
            res["malloc_env"] = MALLOC_ENV
            return res
    raise RuntimeError("cpu baseline leg failed: %s" % out.stderr[-400:])


def bench_summary(res):
    """The numbers a reader of the LAST 2 KB of the line needs (the driver keeps only the tail of stdout): every figure below is a copy of an
    entry further up the same line, never a new measurement.  Must stay <= 1.5 KB; tests/test_bench_launch.py parses it back from the tail."""
    def g(*path, default=None):
        o = res
        for k in path:
            if not isinstance(o, dict) or k not in o:
                return default
            o = o[k]
        return o

    def row(*path):
        r = g(*path)
        if not isinstance(r, dict):
            return None
        if "error" in r:
            return {"error": str(r["error"])[:60]}
        out = {"clips_per_s": r.get("clips_per_s", r.get("value"))}
        if "frac_of_f32_mfma_peak" in r:
            out["frac"] = r["frac_of_f32_mfma_peak"]
        return out

    also = "also"
    s = {
        "value": res.get("value"), "ms_per_step": res.get("ms_per_step"),
        "serial_value": g("serial_value", "value"), "sustained": g("sustained", "value"),
        "roofline_frac": g("roofline", "frac"), "back_to_back_frac": g("roofline", "back_to_back", "frac"),
        "trunk_ms": g("roofline", "back_to_back", "trunk_ms"), "traffic_over_algorithmic": None,
        "split_bf16": None,
        "config2": row(also, "config2_T8_P96_act"), "config3": row(also, "config3_T16_P128_act"), "config4": row(also, "config4_T8_P128_sth_tsm"),
        "config5_f16": None,
        "sth_shipped": {"hot_path": g(also, "sth_shipped_T8_12_P144", "hot_path", "clips_per_s"),
                        "full_forward": g(also, "sth_shipped_T8_12_P144", "full_forward_from_uint8", "value")},
        "glancer": {"ms": g("next_rows", "f2_glancer_mobilenetv2", "ms"), "frac_hbm": g("next_rows", "f2_glancer_mobilenetv2", "frac"),
                    "frac_f32_mfma": g("next_rows", "f2_glancer_mobilenetv2", "frac_of_f32_mfma_peak"),
                    "traffic_over_block_bytes": g("next_rows", "f2_glancer_mobilenetv2", "traffic_over_block_bytes")},
        "policy_ms": g("next_rows", "f2_policy", "ms"), "ingest_frac_hbm": g("next_rows", "f1_ingest_u8", "frac"),
        "full_forward": g("next_rows", "full_forward_from_uint8", "value"),
        "full_forward_pipelined": g("next_rows", "full_forward_from_uint8_pipelined", "value"),
        "evaluate_loop": g("next_rows", "evaluate_loop", "value"),
        "latency_B1_T8_P96_ms": g(also, "latency_small_batch", "B1_T8_P96", "eager_ms"),
        "gather_frac_hbm": g("gather", "frac"),
        "cpu_baseline": g("cpu_baseline", "value"), "gpu_over_cpu": None,
    }
    tr, ab = g("roofline", "traffic"), g("roofline", "algorithmic_bytes_per_launch")
    if tr and ab:
        s["traffic_over_algorithmic"] = round(tr / ab, 2)
    sp = g(also, "split_bf16")
    if isinstance(sp, dict):
        s["split_bf16"] = {"error": str(sp["error"])[:60]} if "error" in sp else {
            "clips_per_s": sp.get("clips_per_s"), "serial": g(also, "split_bf16", "serial", "clips_per_s"),
            "frac_bf16_pipe": g(also, "split_bf16", "roofline", "frac"), "max_abs_logit_diff": sp.get("max_abs_logit_diff_vs_f32")}
    c5 = g(also, "config5_T16_P144_efficientnet_b3", "f16_storage")
    if isinstance(c5, dict):
        s["config5_f16"] = {"clips_per_s": c5.get("clips_per_s"), "local_cnn_ms": g(also, "config5_T16_P144_efficientnet_b3", "f16_storage", "local_cnn", "ms"),
                            "frac_hbm": g(also, "config5_T16_P144_efficientnet_b3", "f16_storage", "local_cnn", "frac"),
                            "frac_structural": g(also, "config5_T16_P144_efficientnet_b3", "f16_storage", "local_cnn", "structural_frac")}
    if s["cpu_baseline"] and res.get("value"):
        s["gpu_over_cpu"] = round(res["value"] / s["cpu_baseline"], 1)
    return s


def _lib_opt(key):
    from adafocus_amd import _lib
    return _lib.get_option(key)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` from a bare shell: become `torch.distributed.run` with N local ranks."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


class DryModel:
    """Stand-in for the GPU step in --dry-run: a deterministic CPU function with the hot path's output shape, so the
    launcher, the collective and the timing protocol can be tested in a container without a GPU."""

    def __init__(self, b, t, classes=200):
        self.w = torch.linspace(-1, 1, 64 * classes).view(64, classes)
        self.b, self.t = b, t

    def hot_path(self, frames, gvec, actions, b, t):
        last = torch.tanh(frames.view(b, -1)[:, :64] @ self.w)
        return last.repeat_interleave(t, 0), last, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--patch", type=int, default=96)
    ap.add_argument("--cpu-baseline", type=int, default=1, help="0 = skip the CPU baseline leg")
    ap.add_argument("--cpu-clips", type=int, default=None, help="(deprecated) 0 = skip the CPU baseline leg")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--profile-steps", type=int, default=5, help="per-launch HIP-event passes for the roofline (median per launch)")
    ap.add_argument("--sustained-steps", type=int, default=240, help="extra soak after the timed region (>= 3 s); 0 = skip")
    ap.add_argument("--full", action="store_true", help="also time the full forward (glancer + policy + hot path)")
    ap.add_argument("--math", choices=["f32", "split_bf16"], default="f32",
                    help="conv arithmetic: f32 = fp32 matrix pipe (default, the reported configuration); split_bf16 = opt-in, "
                         "fp32 operands as three exact bf16 parts on the bf16 matrix pipe, fp32 accumulate")
    ap.add_argument("--tiles", type=str, default="", help="comma list of per-conv tile overrides (tuning)")
    ap.add_argument("--fuse", type=int, default=-1, help="trunk layer fusion: -1 = library default, 0 = off, 1 = on")
    ap.add_argument("--skip-extras", action="store_true", help="only the timed steps + roofline pass (for rocprofv3 runs)")
    ap.add_argument("--streams", type=int, default=3, help="HIP streams the steps are round-robined over (batch i+1's "
                    "trunk overlaps batch i's latency-bound GRU scan); 1 = strictly serial steps")
    ap.add_argument("--dry-run", action="store_true", help="CPU/gloo stand-in for the step: tests the launcher and the protocol")
    ap.add_argument("--cpu-worker", type=str, default="", help="(internal) one process of the all-cores CPU baseline row")
    ap.add_argument("--cpu-baseline-leg", type=str, default="", help="(internal) T,P,threads: the CPU baseline leg, run as its own process")
    ap.add_argument("--inject-init-failure", action="store_true", help="(tests) fail inside the distributed-init guard to exercise the error line")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks run the real step on GPU 0 and gather over gloo (RCCL refuses "
                    "two ranks on one device): exercises the N > 1 path on a single-GPU box; not a scaling measurement")
    a = ap.parse_args()
    if a.cpu_clips is not None and a.cpu_clips <= 0:
        a.cpu_baseline = 0
    if a.cpu_worker:
        cpu_worker(a.cpu_worker)
        return
    if a.cpu_baseline_leg:
        from adafocus_amd import synth
        from tests.helpers import manifest
        tt, pp, thr = (int(v) for v in a.cpu_baseline_leg.split(","))
        torch.set_num_interop_threads(1)
        sd_cpu = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(manifest()["ACT"], 1007).items()}
        print(json.dumps(cpu_baseline_leg(sd_cpu, tt, pp, thr)), flush=True)
        return

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if not a.dry_run and not a.share_gpu:
            # the likeliest failure of a first unattended multi-GPU run: the box exposes fewer devices than --gpus.  Say so in ONE
            # line (the init-failure form) from the parent, before N ranks each die with a traceback in torch.cuda.set_device
            have = torch.cuda.device_count() if torch.cuda.is_available() else 0
            if have < a.gpus:
                init_failure_line(a, a.gpus, 0, "nccl", RuntimeError(
                    "--gpus %d but this process sees %d HIP device(s) (HIP_VISIBLE_DEVICES=%r, ROCR_VISIBLE_DEVICES=%r)"
                    % (a.gpus, have, os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES"))), launched=False)
                raise SystemExit(3)
        self_launch(a.gpus)                       # does not return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    dry = a.dry_run
    backend = "gloo" if (dry or a.share_gpu) else "nccl"
    if dry:
        dev = torch.device("cpu")
    else:
        if a.share_gpu:
            local = 0
        try:                                       # inside the guard too: a rank whose device does not exist must not die with a bare traceback
            have = torch.cuda.device_count() if torch.cuda.is_available() else 0
            need = 1 if a.share_gpu else max(local + 1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
            if have < need:       # EVERY rank of the node sees the same shortfall, so rank 0 is among those that report it
                raise RuntimeError("rank %d wants HIP device %d (of %d local ranks) but this process sees %d device(s) (HIP_VISIBLE_DEVICES=%r, "
                                   "ROCR_VISIBLE_DEVICES=%r)" % (rank, local, need, have, os.environ.get("HIP_VISIBLE_DEVICES"),
                                                                 os.environ.get("ROCR_VISIBLE_DEVICES")))
            torch.cuda.set_device(local)          # rank i <-> GPU i of this node
        except Exception as exc:      # noqa: BLE001
            init_failure_line(a, world, rank, backend, exc)
            raise SystemExit(3)
        dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            # rendezvous + the FIRST collective inside one guard: an RCCL problem (IPC mode, topology, a missing device) usually shows
            # up at communicator creation, i.e. at the first collective, not at init_process_group
            if a.inject_init_failure:
                raise RuntimeError("injected failure (--inject-init-failure): exercises the error line")
            if backend == "gloo":
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe)
            if not dry:
                torch.cuda.synchronize()
            if int(probe.item()) != world:
                raise RuntimeError("first all_reduce returned %r, expected %d" % (probe.item(), world))
        except Exception as exc:      # noqa: BLE001 -- whatever it is, the driver must get ONE diagnosable line instead of N tracebacks
            init_failure_line(a, world, rank, backend, exc)
            raise SystemExit(3)
    numa_node = None
    if world > 1 and not dry:
        from adafocus_amd.parallel import bind_to_gpu_numa
        numa_node = bind_to_gpu_numa(local)       # the rank's host threads next to its GPU

    def sync():
        if not dry:
            torch.cuda.synchronize()

    from adafocus_amd import synth, workload
    from adafocus_amd.parallel import gather_logits, gather_logits_async
    comm_stream = torch.cuda.Stream(device=dev) if (world > 1 and not dry) else None

    b, t, p = a.batch, a.frames, a.patch
    if dry:
        model, sd, trunk = DryModel(b, t), None, None
        frames = torch.randn(b * t, 3, 8, 8)
        actions = torch.rand(b * t, 2)
        gvec = torch.randn(b, t, 1280)
        streams = [None] * max(a.streams, 1)
    else:
        from adafocus_amd.gfv_net import GFV
        model = GFV(act_args(t, p, b)).eval()
        sd = synth_model_state(model, 1007)
        model.load_state_dict(sd, strict=True)
        model = model.to(dev)
        # synthetic inputs, resident in HBM before the timed region (per-rank shard of the clip set)
        frames = torch.from_numpy(synth.synth_frames(b, t, 224, seed=100 + rank)).to(dev).view(b * t, 3, 224, 224)
        _, act_np = synth.synth_actions(b * t, 7, seed=2 + rank)
        actions = torch.from_numpy(act_np).to(dev)
        gvec = torch.randn((b, t, 1280), device=dev)
        model.focuser.net.set_math(a.math)
        trunk = model.focuser.net._sync()
        if a.tiles:
            trunk.set_tiles([int(v) for v in a.tiles.split(",")])
        if a.fuse >= 0:
            trunk.set_fusion(a.fuse)
        streams = [torch.cuda.Stream(device=dev) for _ in range(max(a.streams, 1))]

    def step(i):
        # consecutive batches are independent: enqueue them on alternating streams (software pipelining of
        # the eval loop); every step still does all of its work, and the timed region ends with a device sync
        if dry:
            last = model.hot_path(frames, gvec, actions, b, t)[1]
            return gather_logits(last) if world > 1 else last
        with torch.no_grad(), torch.cuda.stream(streams[i % len(streams)]):
            logits, last, _ = model.hot_path(frames, gvec, actions, b, t)
            if world > 1:      # the logits all-gather on its own stream, ordered after this step's scan; the next trunk does not wait for it
                last = gather_logits_async(last, comm_stream)
        return last

    def timed_steps(n):
        sync()
        if world > 1:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        out = None
        for i in range(n):
            out = step(i)
        sync()
        if world > 1:
            dist.barrier()
        sync()
        mine = time.perf_counter() - t0
        every = [mine]
        if world > 1:
            tt = torch.tensor([mine], device=dev, dtype=torch.float64)
            allt = [torch.zeros_like(tt) for _ in range(world)]
            dist.all_gather(allt, tt)
            every = [float(x.item()) for x in allt]
        return max(every), every, out

    for i in range(a.warmup):
        step(i)
    elapsed, per_rank_s, out = timed_steps(a.steps)
    assert torch.isfinite(out).all()
    if world > 1:
        assert out.shape[0] == world * b, out.shape      # the gathered logits of every rank's shard

    value = b * world * a.steps / elapsed
    res = {
        "metric": "clips/sec (T=%d, patch=%d^2, ResNet-50 local)" % (t, p), "value": round(value, 2), "unit": "clips/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if a.math == "f32" else "f32 (operands split into 3 bf16 parts, 6 bf16-MFMA products, f32 accumulate)",
        "data": "synthetic",
        "config": {"workload": "ActivityNet AdaFocus hot path: gather + ResNet-50 local CNN + GRU classifier, "
                               "T=%d, P=%d, B=%d clips/GPU (%d patches/GPU/step), random-init weights seed 1007"
                               % (t, p, b, b * t),
                   "global_batch": b * world, "frames": t, "patch": p, "parallelism": "dp%d" % world,
                   "streams": len(streams)},
        "ranks": world, "backend": ("gloo" if dry else "gloo (all ranks share GPU 0)" if a.share_gpu else "nccl (RCCL)") if world > 1 else None,
        # ranks of an initialised RCCL ("nccl") process group -- taken from torch.distributed, never inferred
        "rccl_ranks": (dist.get_world_size() if (dist is not None and dist.is_initialized() and dist.get_backend() == "nccl") else 0),
        "numa_node_rank0": numa_node,
        "per_rank_clips_per_s": [round(b * a.steps / s, 1) for s in per_rank_s],
        "gflop_per_clip": round(workload.hot_path_flops_per_clip(t, p) / 1e9, 2),
    }
    if dry:
        res["dry_run"] = True

    if a.sustained_steps > 0 and not a.skip_extras:
        # the same step for >= 3 s: the 30-step figure above is a ~0.4 s sample
        s_el, _, _ = timed_steps(a.sustained_steps)
        res["sustained"] = {"value": round(b * world * a.sustained_steps / s_el, 2), "unit": "clips/s", "steps": a.sustained_steps,
                            "seconds": round(s_el, 3), "ms_per_step": round(1e3 * s_el / a.sustained_steps, 3)}

    if not dry and len(streams) > 1 and not a.skip_extras:
        # the same steps strictly serial (one stream): `value` overlaps batch i+1's trunk with batch i's latency-bound GRU scan
        # and launch tails across %d streams, each with its own trunk workspace; this is what a single stream gets
        saved = streams
        streams = streams[:1]
        for i in range(3):
            step(i)
        s1, _, _ = timed_steps(a.steps)
        streams = saved
        res["serial_value"] = {"value": round(b * world * a.steps / s1, 2), "unit": "clips/s", "streams": 1, "steps": a.steps,
                               "ms_per_step": round(1e3 * s1 / a.steps, 3),
                               "note": "`value` uses %d streams (and %d trunk workspaces of %.1f GB); this is the strictly serial rate" % (
                                   len(saved), len(saved), (trunk._lib.adaf_resnet50_workspace_bytes(trunk._net, b * t, p) / 1e9) if trunk is not None else 0.0)}

    if rank == 0 and not dry:
        # ---- roofline of the dominant kernel: per-launch HIP events on the launch stream -------------
        conv_ms = conv_fl = tot_ms = conv_by = 0.0
        nconv = 0
        from adafocus_amd.utils import get_patch_nhwc4
        x4 = get_patch_nhwc4(frames, actions, p)
        trunk.profile(x4)                                # untimed: the first bracketed pass pays for event creation
        nps = max(a.profile_steps, 1)
        runs = [trunk.profile(x4) for _ in range(nps)]
        per_launch = []
        for i, e in enumerate(runs[0]):                  # per launch: the median over the passes
            ms = statistics.median(r[i]["ms"] for r in runs)
            per_launch.append(dict(e, ms=ms))
            tot_ms += ms * nps
            if e["flops"] > 0:
                conv_ms += ms * nps
                conv_fl += e["flops"] * nps
                conv_by += e.get("bytes", 0.0) * nps
                nconv += nps
        achieved = conv_fl / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        # the same launches back to back (two events around three whole trunk passes): what the event brackets of the
        # per-launch pass add between kernels is not kernel time
        with torch.no_grad():
            trunk.forward(x4)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                trunk.forward(x4)
            e1.record()
            torch.cuda.synchronize()
        wall_ms = e0.elapsed_time(e1) / 3
        # (nothing is subtracted: the max-pool rides in the stem's launch and, since round 3, the average pool in the last conv3's
        # epilogue -- the pass IS the conv engine's launches)
        b2b = (conv_fl / nps) / (wall_ms * 1e-3) / 1e12
        res["roofline"] = {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                           "unit": "TFLOP/s", "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4), "traffic": load_traffic(t, p, b),
                           "kernel": "conv engine (implicit-GEMM conv+BN+ReLU on v_mfma_f32_32x32x2_f32: conv_gemm_glds_kernel, "
                                     "the fused stage-1 / stem kernels), %d launches/step" % (nconv // nps),
                           "avg_launch_ms": round(conv_ms / max(nconv, 1), 4),
                           "flop_per_launch": round(conv_fl / max(nconv, 1), 1),
                           "algorithmic_bytes_per_launch": round(conv_by / max(nconv, 1), 1),
                           "trunk_ms_per_step": round(tot_ms / nps, 3),
                           "back_to_back": {"trunk_ms": round(wall_ms, 3), "achieved": round(b2b, 2),
                                            "frac": round(b2b / MFMA_F32_PEAK_TFLOPS, 4),
                                            "note": "two events around whole trunk passes (both pools ride in conv launches); achieved/frac "
                                                    "above are from per-launch event brackets (conservative)"}}
        # the gather, priced against HBM
        for _ in range(3):          # let the caching allocator settle on this stream before timing
            get_patch_nhwc4(frames, actions, p)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(20):
            get_patch_nhwc4(frames, actions, p)
        ev1.record()
        torch.cuda.synchronize()
        crop_ms = ev0.elapsed_time(ev1) / 20
        crop_bytes = float(workload.crop_bytes_per_patch(p)) * b * t
        res["gather"] = {"bound": "hbm", "achieved": round(crop_bytes / (crop_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(crop_bytes / (crop_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "ms": round(crop_ms, 4), "bytes_per_patch": workload.crop_bytes_per_patch(p),
                         "note": "adaf_crop_gather_f32 as its own launch (get_patch for callers that want the patch tensor); in the timed step "
                                 "the gather rides in the trunk's first launch (adaf_resnet50_forward_frames: the stem fetches its windows "
                                 "from the frames), so this launch and the patch tensor do not exist there"}
        def extra(group, key, fn):      # an `also` / `next_rows` entry must never fail the bench
            try:
                res.setdefault(group, {})[key] = fn()
            except Exception as exc:
                res.setdefault(group, {})[key] = {"error": repr(exc)[:300]}
        if not a.skip_extras and world == 1:
            # the other BASELINE configurations, hot path per GPU, same protocol as `value` (>= 30 steps, >= 1 s for the short one)
            extra("also", "config2_T8_P96_act", lambda: X.act_hot_path_row(dev, 8, 96, b, streams, 120))
            extra("also", "config3_T16_P128_act", lambda: X.act_hot_path_row(dev, 16, 128, b, streams, 40))
            extra("also", "config4_T8_P128_sth_tsm", lambda: X.sth_hot_path_row(dev, b, streams, 60))
            extra("also", "latency_small_batch", lambda: X.latency_rows(dev))
            res["gather_resize"] = None
            try:
                res["gather_resize"] = X.gather_resize_row(dev, frames, p)
            except Exception as exc:
                res["gather_resize"] = {"error": repr(exc)[:300]}
        if a.math == "f32" and not a.skip_extras and world == 1:   # (step() holds a collective when world > 1)
            # opt-in arithmetic (not the reported configuration): same step with the convs on the bf16 matrix pipe, priced on ITS pipe
            extra("also", "split_bf16", lambda: X.split_bf16_row(dev, model, frames, gvec, actions, b, t, p, streams, a.steps, step))
            if "error" not in res["also"]["split_bf16"]:
                res["also"]["opt_in_split_bf16"] = {"clips_per_s": res["also"]["split_bf16"]["clips_per_s"],
                                                    "max_abs_logit_diff_vs_f32": res["also"]["split_bf16"]["max_abs_logit_diff_vs_f32"],
                                                    "note": "(round-4 key, kept for comparison: see also.split_bf16)"}
            model.focuser.net.set_math("f32")
        if world == 1 and not a.skip_extras:
            # ---- rows f1/f2 of the scope table, measured the same way (inputs resident, HIP events): uint8 ingest,
            # glancer, policy, and the whole forward from the loader's uint8 clips
            try:
                from adafocus_amd.transforms import ingest_uint8

                def timed(fn, iters=5):
                    for _ in range(3):
                        fn()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(iters):
                        out = fn()
                    e1.record()
                    torch.cuda.synchronize()
                    return e0.elapsed_time(e1) / iters, out

                u8 = torch.randint(0, 256, (b, 224, 224, t * 3), device=dev, dtype=torch.uint8)
                with torch.no_grad():
                    ing_ms, fr4 = timed(lambda: ingest_uint8(u8, t))
                    gl_ms, (fmap, fvec) = timed(lambda: model.glancer.net.features_from_nhwc4(fr4), 3)
                    table = model.focuser.action_table(dev)
                    pol_ms, _ = timed(lambda: model.focuser.policy.policy_old.act_sequence_nhwc(fmap, b, t, table))
                    full_ms, _ = timed(lambda: model.offline_forward_nhwc4(ingest_uint8(u8, t), b, t), 3)
                    # consecutive batches pipelined inside the model (its own streams + events), as evaluate.validate runs it
                    u8s = [u8, u8.clone(), u8.clone()]
                    for i in range(3):
                        model.offline_forward_pipelined(u8s[i % 3], t)
                    model.pipeline_flush()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for i in range(9):
                        model.offline_forward_pipelined(u8s[i % 3], t)
                    model.pipeline_flush()
                    torch.cuda.synchronize()
                    piped_ms = (time.perf_counter() - t1) / 9 * 1e3
                ing_bytes = float(b * t * 224 * 224 * (3 + 16))
                gl_fusion = int(model.glancer.net._engine.fusion)        # adaf_mobilenetv2_set_fusion bits (True = 1)
                gl_bytes = float(b * t) * workload.mobilenetv2_bytes_per_frame(224, fused=bool(gl_fusion & 1), fused_tail=model.glancer.net.fused_tail(),
                                                                                whole_blocks=not (gl_fusion & 8), strips=bool(_lib_opt("mb_strip")))
                gl_flop = 2.0 * workload.mobilenetv2_macs_per_frame(224)
                gl_block = float(b * t) * workload.mobilenetv2_block_bytes_per_frame(224)
                gl_traffic = load_glancer_traffic(b * t)
                res["next_rows"] = {
                    "f1_ingest_u8": {"bound": "hbm", "ms": round(ing_ms, 4), "achieved": round(ing_bytes / ing_ms / 1e6, 1),
                                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ing_bytes / ing_ms / 1e6 / HBM_PEAK_GBS, 4),
                                     "bytes_per_pixel": 19},
                    "f2_glancer_mobilenetv2": {"bound": "hbm", "ms": round(gl_ms, 3), "achieved": round(gl_block / gl_ms / 1e6, 1),
                                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gl_block / gl_ms / 1e6 / HBM_PEAK_GBS, 4),
                                               "bytes_per_frame": int(gl_block / (b * t)),
                                               "bytes_are": "BLOCK-LEVEL algorithmic bytes: frame in, every block-boundary tensor written once + read "
                                                            "once (+ identity), 1280-channel map + pooled vector out, fp32 "
                                                            "(adafocus_amd/workload.py:mobilenetv2_block_bytes_per_frame)",
                                               "plan_bytes_per_frame": int(gl_bytes / (b * t)),
                                               "plan_frac": round(gl_bytes / gl_ms / 1e6 / HBM_PEAK_GBS, 4),
                                               "plan_bytes_are": "activations in + out of every launch of the FUSED plan that runs "
                                                                 "(workload.mobilenetv2_bytes_per_frame; the round-1..4 denominator)",
                                               "traffic_bytes_per_frame": gl_traffic,
                                               "traffic_over_block_bytes": round(gl_traffic / (gl_block / (b * t)), 2) if gl_traffic else None,
                                               "gflop_per_frame": round(gl_flop / 1e9, 3), "tflops": round(gl_flop * b * t / gl_ms / 1e9, 1),
                                               "frac_of_f32_mfma_peak": round(gl_flop * b * t / gl_ms / 1e9 / MFMA_F32_PEAK_TFLOPS, 4)},
                    "f2_policy": {"ms": round(pol_ms, 3), "note": "1x1 conv + FC over all B*T frames, GRU scan over T, arg-max + grid lookup"},
                    "full_forward_from_uint8": {"value": round(b / full_ms * 1e3, 1), "unit": "clips/s", "ms": round(full_ms, 3),
                                                "note": "ingest + glancer + policy + hot path (GFV.offline_forward_nhwc4), serial on one stream"},
                    "full_forward_from_uint8_pipelined": {"value": round(b / piped_ms * 1e3, 1), "unit": "clips/s", "ms": round(piped_ms, 3),
                                                          "note": "GFV.offline_forward_pipelined: batch i+1's ingest + glancer + policy on the "
                                                                  "model's front stream while batch i's hot path runs on its back stream"},
                }
                del u8, u8s, fr4, fmap, fvec
            except Exception as exc:  # upstream of the timed path; never fail the bench on it
                res["next_rows"] = {"error": repr(exc)[:300]}
        if world == 1 and not a.skip_extras:
            eargs = act_args(t, p, b)
            extra("next_rows", "evaluate_loop", lambda: X.evaluate_loop_row(dev, model, eargs, b, t))
            extra("also", "validate_sth_loop_T8_P128", lambda: X.validate_sth_row(dev, b))
            extra("also", "sth_shipped_T8_12_P144", lambda: X.sth_shipped_row(dev, b, streams))
        if world == 1 and not a.skip_extras and (t, p) == (16, 96):
            try:
                res.setdefault("also", {})["config5_T16_P144_efficientnet_b3"] = config5_row(dev, b, streams, frames)
            except Exception as exc:  # never fail the bench on an `also` row
                res.setdefault("also", {})["config5_T16_P144_efficientnet_b3"] = {"error": repr(exc)[:300]}
        if os.environ.get("ADAF_BENCH_LAUNCHES"):
            res["launches"] = [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in e.items()} for e in per_launch]
        if a.full and not a.skip_extras:
            try:
                scan = frames.view(b, t * 3, 224, 224)
                with torch.no_grad():
                    for _ in range(2):
                        model.offline_forward(scan, scan)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(3):
                        model.offline_forward(scan, scan)
                    torch.cuda.synchronize()
                res["full_forward"] = {"value": round(3 * b / (time.perf_counter() - t1), 2), "unit": "clips/s",
                                       "note": "GFV.offline_forward: glancer (adaf_mobilenetv2) + policy on the engine + hot path, 3 iterations"}
            except Exception as exc:  # upstream of the timed path; never fail the bench on it
                res["full_forward"] = {"error": repr(exc)[:200]}
        if world == 1 and a.cpu_baseline and not a.skip_extras:
            res["cpu_baseline"] = cpu_baseline(t, p, a.cpu_threads)
    if rank == 0:
        res.pop("summary", None)
        res["summary"] = bench_summary(res)      # LAST key: the driver keeps the tail of the line
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
