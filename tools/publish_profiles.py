#!/usr/bin/env python3
"""Copy the summaries tools/profile_bench.sh left in gpurun_out/ into profiles/ and derive profiles/<tag>_traffic.json
(the per-launch HBM bytes bench.py reports as roofline.traffic) from the FETCH_SIZE / WRITE_SIZE passes.
usage: publish_profiles.py [tag=r2]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")
NAMES = {"trace1": "bench_kernel_trace", "trace3": "bench_kernel_trace_3streams", "mfma": "bench_pmc_mfma",
         "fetch": "bench_pmc_fetch_size", "write": "bench_pmc_write_size"}
CMDS = {
    "trace1": "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 20 --warmup 4 --skip-extras --streams 1 --cpu-baseline 0",
    "trace3": "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 30 --warmup 6 --skip-extras --cpu-baseline 0",
    "mfma": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -- python bench.py --steps 4 --warmup 2 --skip-extras --profile-steps 1 --streams 1 --cpu-baseline 0",
    "fetch": "rocprofv3 --pmc FETCH_SIZE --output-format csv -- python bench.py --steps 4 --warmup 2 --skip-extras --profile-steps 1 --streams 1 --cpu-baseline 0",
    "write": "rocprofv3 --pmc WRITE_SIZE --output-format csv -- python bench.py --steps 4 --warmup 2 --skip-extras --profile-steps 1 --streams 1 --cpu-baseline 0",
}
CONV = re.compile(r"conv_gemm_|conv_fused_tail_|stem7x7_")


def counter_rows(path, counter):
    """(dispatches, sum) over the conv-engine kernels of one summary; stem dispatches = trunk passes"""
    disp, total, passes = 0, 0.0, 0
    for line in open(path):
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        if len(cells) < 5 or cells[1] != counter or not CONV.search(cells[0]):
            continue
        disp += int(cells[2])
        total += float(cells[3])
        if "stem7x7_" in cells[0]:
            passes += int(cells[2])
    return disp, total, passes


EXTRA = {   # round 3: summaries that are copied as they are (tools/profile_r3.sh)
    "effnet_f16_trace": "rocprofv3 --kernel-trace --stats -- python tools/effnet_probe.py 1024 144 5 f16   (EfficientNet-B3 local CNN, fp16 storage, 3 warm-up + 5 timed forwards)",
    "effnet_f32_trace": "rocprofv3 --kernel-trace --stats -- python tools/effnet_probe.py 1024 144 5 f32   (fp32 storage)",
    "effnet_f16_sq": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -- python tools/effnet_probe.py 1024 144 5 f16",
    "effnet_f16_fetch": "rocprofv3 --pmc FETCH_SIZE -- python tools/effnet_probe.py 1024 144 5 f16   (KB; x2 for wide coalesced reads on gfx950, MI355X_MICROARCH.md)",
    "effnet_f16_write": "rocprofv3 --pmc WRITE_SIZE -- python tools/effnet_probe.py 1024 144 5 f16   (KB)",
    "resize_trace": "rocprofv3 --kernel-trace --stats -- python tools/crop_resize_probe.py 5   (adaf_crop_resize_f32, 1024 frames, S = 128 / 192 / mixed -> 96^2)",
    "resize_fetch": "rocprofv3 --pmc FETCH_SIZE -- python tools/crop_resize_probe.py 5   (KB; x2 for wide coalesced reads)",
    "resize_write": "rocprofv3 --pmc WRITE_SIZE -- python tools/crop_resize_probe.py 5   (KB)",
}
EXTRA_R4 = {   # round 4 (tools/profile_r4.sh)
    "fullfwd_trace_serial": "rocprofv3 --kernel-trace -- python tools/fullfwd_probe.py serial 10, condensed by tools/overlap_report.py   (full forward from uint8 clips, B = 64, T = 16, P = 96: ingest + glancer + policy + hot path on ONE stream)",
    "fullfwd_trace_2streams": "rocprofv3 --kernel-trace -- python tools/fullfwd_probe.py two 10, condensed by tools/overlap_report.py   (the same work through GFV.offline_forward_pipelined: front half of batch i+1 and back half of batch i on the model's two streams)",
    "glancer_trace": "rocprofv3 --kernel-trace --stats -- python tools/glancer_probe.py 1024   (MobileNetV2 glancer, 1024 frames of 224^2, 2 warm-up + 3 timed forwards)",
    "glancer_fetch": "rocprofv3 --pmc FETCH_SIZE -- python tools/glancer_probe.py 1024   (KB; x2 for wide coalesced reads on gfx950, MI355X_MICROARCH.md)",
    "glancer_write": "rocprofv3 --pmc WRITE_SIZE -- python tools/glancer_probe.py 1024   (KB)",
    "effnet_f16_trace": "rocprofv3 --kernel-trace --stats -- python tools/effnet_probe.py 1024 144 5 f16   (EfficientNet-B3 local CNN, fp16 storage, whole-block MBConv kernels on; 3 warm-up + 5 timed forwards)",
    "effnet_f16_sq": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -- python tools/effnet_probe.py 1024 144 5 f16",
    "effnet_f16_fetch": "rocprofv3 --pmc FETCH_SIZE -- python tools/effnet_probe.py 1024 144 5 f16   (KB; x2 for wide coalesced reads on gfx950)",
    "effnet_f16_write": "rocprofv3 --pmc WRITE_SIZE -- python tools/effnet_probe.py 1024 144 5 f16   (KB)",
}
EXTRA_R5 = {   # round 5 (tools/profile_r5.sh)
    "split_trace": "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 20 --warmup 4 --skip-extras --streams 1 --cpu-baseline 0 --math split_bf16   (the opt-in arithmetic: also.split_bf16)",
    "split_mfma": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -- python bench.py --steps 4 --warmup 2 --skip-extras --profile-steps 1 --streams 1 --cpu-baseline 0 --math split_bf16",
    # tools/profile_r5_effnet.sh (config 5 after the zero-scratch whole-block kernels)
    "effnet_f16_trace": "rocprofv3 --kernel-trace --stats -- python tools/effnet_probe.py 1024 144 5 f16   (EfficientNet-B3 local CNN, fp16 storage; 3 warm-up + 5 timed forwards)",
    "effnet_f16_sq": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -- python tools/effnet_probe.py 1024 144 5 f16",
    "effnet_f16_icache": "rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES -- python tools/effnet_probe.py 1024 144 5 f16",
    "effnet_f16_fetch": "rocprofv3 --pmc FETCH_SIZE -- python tools/effnet_probe.py 1024 144 5 f16   (KB; x2 for wide coalesced reads on gfx950)",
    "effnet_f16_write": "rocprofv3 --pmc WRITE_SIZE -- python tools/effnet_probe.py 1024 144 5 f16   (KB)",
}
EXTRA_R6 = {   # round 6 (tools/profile_r6.sh; the full-forward traces: tools/exp/r6_fullfwd.sh)
    "fullfwd_trace_serial": "rocprofv3 --kernel-trace -- python tools/fullfwd_probe.py serial 10, condensed by tools/overlap_report.py   (full forward from uint8 clips, B = 64, T = 16, P = 96: ingest + glancer + policy + hot path on ONE stream; 20.9 ms per batch = 3.06 k clips/s on that box)",
    "fullfwd_trace_2streams": "rocprofv3 --kernel-trace -- python tools/fullfwd_probe.py two 10, condensed by tools/overlap_report.py   (the same work through GFV.offline_forward_pipelined; 21.1 ms)",
    "glancer_trace": "rocprofv3 --kernel-trace --stats -- python tools/glancer_probe.py 1024   (MobileNetV2 glancer, 1024 frames of 224^2, strip-walking front kernels of csrc/mbstrip.hip; 2 warm-up + 3 timed forwards; two 512-frame chunks side by side on two streams)",
    "glancer_trace_serial": "rocprofv3 --kernel-trace --stats -- python tools/glancer_probe.py 1024 5   (the same forward with chunk pairing off: every kernel alone on the device)",
    "glancer_fetch": "rocprofv3 --pmc FETCH_SIZE -- python tools/glancer_probe.py 1024   (KB; x2 for wide coalesced reads on gfx950, MI355X_MICROARCH.md)",
    "glancer_write": "rocprofv3 --pmc WRITE_SIZE -- python tools/glancer_probe.py 1024   (KB)",
    "glancer_sq": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE -- python tools/glancer_probe.py 1024 5",
    "split_trace": "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 20 --warmup 4 --skip-extras --streams 1 --cpu-baseline 0 --math split_bf16   (the opt-in arithmetic: also.split_bf16)",
    "split_mfma": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -- python bench.py --steps 4 --warmup 2 --skip-extras --profile-steps 1 --streams 1 --cpu-baseline 0 --math split_bf16",
    "split_lds": "rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_INSTS_MFMA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -- python bench.py --steps 4 --warmup 2 --skip-extras --profile-steps 1 --streams 1 --cpu-baseline 0 --math split_bf16",
}
if tag >= "r4":
    EXTRA = EXTRA_R4
if tag >= "r5":
    EXTRA = EXTRA_R5
if tag >= "r6":
    EXTRA = EXTRA_R6


def all_kernels_sum(path, counter, once_per_forward):
    """(sum of `counter` over every kernel of a summary, dispatches of the kernel that runs once per forward)"""
    total, fwd = 0.0, 0
    for line in open(path):
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        if len(cells) < 5 or cells[1] != counter:
            continue
        total += float(cells[3])
        if once_per_forward in cells[0]:
            fwd += int(cells[2])
    return total, fwd


for short, cmd in EXTRA.items():
    src = os.path.join(OUT, "%s_%s.md" % (tag, short))
    if os.path.exists(src):
        body = open(src).read().split("\n", 1)[1]
        open(os.path.join(PROF, "%s_%s.md" % (tag, short)), "w").write("# %s\n%s" % (cmd, body))
        print("published", short)

for short, name in NAMES.items():
    src = os.path.join(OUT, "%s_%s.md" % (tag, short))
    if not os.path.exists(src):
        print("missing", src)
        continue
    body = open(src).read().split("\n", 1)[1]
    open(os.path.join(PROF, "%s_%s.md" % (tag, name)), "w").write("# %s\n%s" % (CMDS[short], body))
    print("published", name)

fd, fs, passes = counter_rows(os.path.join(PROF, "%s_bench_pmc_fetch_size.md" % tag), "FETCH_SIZE")
wd, ws, _ = counter_rows(os.path.join(PROF, "%s_bench_pmc_write_size.md" % tag), "WRITE_SIZE")
per_launch = int((2 * fs + ws) * 1024 / fd)
json.dump({
    "source": "profiles/%s_bench_pmc_fetch_size.md + %s_bench_pmc_write_size.md (rocprofv3 --pmc, separate passes; conv_gemm_*, "
              "conv_fused_tail_* and stem7x7_* kernels)" % (tag, tag),
    "workload": {"frames": 16, "patch": 96, "batch": 64},
    "conv_dispatches_counted": [fd, wd],
    "trunk_conv_launches_per_step": fd // max(passes, 1),
    "passes": passes,
    "fetch_kb_sum": fs, "write_kb_sum": ws,
    "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE under-counts wide coalesced reads by 2x on gfx950, MI355X_MICROARCH.md)",
    "hbm_bytes_per_launch": per_launch,
    "algorithmic_bytes_note": "sum over the launches of in + out (+ residual) + weights, bench.py launch table: see roofline.flop_per_launch / DESIGN 3.2",
}, open(os.path.join(PROF, "%s_traffic.json" % tag), "w"), indent=1)
print("traffic: %d dispatches over %d passes, %.1f MB / launch" % (fd, passes, per_launch / 1e6))

# EfficientNet-B3 (config 5): HBM bytes per patch of the whole forward, every kernel counted
ef_f, ef_w = os.path.join(PROF, "%s_effnet_f16_fetch.md" % tag), os.path.join(PROF, "%s_effnet_f16_write.md" % tag)
if os.path.exists(ef_f) and os.path.exists(ef_w):
    fs, nf = all_kernels_sum(ef_f, "FETCH_SIZE", "ef_stem_")
    ws, nw = all_kernels_sum(ef_w, "WRITE_SIZE", "ef_stem_")
    per_patch = int((2 * fs / max(nf, 1) + ws / max(nw, 1)) * 1024 / 1024)
    json.dump({"source": "profiles/%s_effnet_f16_fetch.md + %s_effnet_f16_write.md (rocprofv3 --pmc, separate passes, EVERY kernel of the forward)" % (tag, tag),
               "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per forward / 1024 patches (FETCH_SIZE under-counts wide coalesced reads by 2x on gfx950)",
               "f16": {"patches": 1024, "P": 144, "forwards_counted": [nf, nw], "fetch_kb_per_forward": fs / max(nf, 1), "write_kb_per_forward": ws / max(nw, 1),
                       "bytes_per_patch": per_patch}},
              open(os.path.join(PROF, "%s_effnet_traffic.json" % tag), "w"), indent=1)
    print("effnet traffic: %.2f MB / patch over %d forwards" % (per_patch / 1e6, nf))
gl_f, gl_w = os.path.join(PROF, "%s_glancer_fetch.md" % tag), os.path.join(PROF, "%s_glancer_write.md" % tag)
if os.path.exists(gl_f) and os.path.exists(gl_w):
    fs, nf = all_kernels_sum(gl_f, "FETCH_SIZE", "mb_stem_b1")       # (matches mb_stem_b1_w_kernel and, round 6, mb_stem_b1_s_kernel)
    ws, nw = all_kernels_sum(gl_w, "WRITE_SIZE", "mb_stem_b1")
    # (the stem + block-1 kernel runs once per 512-frame chunk: two dispatches per 1024-frame forward)
    per_frame = int((2 * fs / max(nf, 1) + ws / max(nw, 1)) * 1024 / 512)
    json.dump({"source": "profiles/%s_glancer_fetch.md + %s_glancer_write.md" % (tag, tag), "frames": 1024, "chunks_counted": [nf, nw],
               "bytes_per_frame": per_frame}, open(os.path.join(PROF, "%s_glancer_traffic.json" % tag), "w"), indent=1)
    print("glancer traffic: %.2f MB / frame" % (per_frame / 1e6))
