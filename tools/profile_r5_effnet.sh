#!/bin/bash
# Round-5 rocprofv3 evidence for BASELINE config 5 (EfficientNet-B3 local CNN, 1024 x 144^2, fp16 storage): kernel trace, SQ occupancy /
# VALU counters, instruction-cache counters, FETCH_SIZE and WRITE_SIZE -- one counter group per run, kernel-trace / stats domains only.
# Summaries land in gpurun_out/r5_effnet_*.md; `python tools/publish_profiles.py r5` copies them into profiles/ and derives
# profiles/r5_effnet_traffic.json.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; rocprofv3 "$@" > $OUT/r5_$name.log 2>&1; }
# (effnet_plan 255 = without ADAF_EF_PLAN_PAIR_CHUNKS: one chunk on one stream, every kernel alone on the device -- the plan the committed r5_effnet_* summaries
#  were collected under, before the pairing existed; the default forward runs the same kernels as two half chunks side by side)
EFF16="python $R/tools/effnet_probe.py 1024 144 5 f16 255"
run effnet_f16_trace --kernel-trace --stats --output-format csv -d $OUT/prof_r5_effnet_f16_trace -- $EFF16
run effnet_f16_sq --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/prof_r5_effnet_f16_sq -- $EFF16
run effnet_f16_icache --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES --output-format csv -d $OUT/prof_r5_effnet_f16_icache -- $EFF16
run effnet_f16_fetch --pmc FETCH_SIZE --output-format csv -d $OUT/prof_r5_effnet_f16_fetch -- $EFF16
run effnet_f16_write --pmc WRITE_SIZE --output-format csv -d $OUT/prof_r5_effnet_f16_write -- $EFF16
cd $R
for d in effnet_f16_trace effnet_f16_sq effnet_f16_icache effnet_f16_fetch effnet_f16_write; do
  python tools/summarize_rocprof.py $OUT/prof_r5_$d $OUT/r5_$d.md "$d" || true
  find $OUT/prof_r5_$d -name '*.csv' -size +1M -delete
done
tail -1 $OUT/r5_effnet_f16_trace.log
ls $OUT | grep r5_effnet
