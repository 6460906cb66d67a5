#!/bin/bash
# Round-3 rocprofv3 evidence (run on the GPU box through gpurun; summaries land in gpurun_out/r3_*.md, then
# `python tools/publish_profiles.py r3` copies them into profiles/):
#   bench:   kernel traces (serial / 3 streams) + MFMA / FETCH_SIZE / WRITE_SIZE passes of the headline step  (tools/profile_bench.sh r3)
#   effnet:  kernel traces of the EfficientNet-B3 local CNN (1024 x 144^2) in fp16 and fp32 storage, SQ + FETCH_SIZE + WRITE_SIZE passes (fp16)
#   resize:  kernel trace + FETCH_SIZE + WRITE_SIZE of the resampling gather (row N1)
# PMC passes run on their own (kernel-trace / stats domains only), one counter group per run.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() {  # name, rocprof args..., -- command
  local name=$1; shift
  rocprofv3 "$@" > $OUT/r3_$name.log 2>&1
}
EFF16="python $R/tools/effnet_probe.py 1024 144 5 f16"
EFF32="python $R/tools/effnet_probe.py 1024 144 5 f32"
RSZ="python $R/tools/crop_resize_probe.py 5"
run effnet_f16_trace --kernel-trace --stats --output-format csv -d $OUT/prof_r3_effnet_f16_trace -- $EFF16
run effnet_f32_trace --kernel-trace --stats --output-format csv -d $OUT/prof_r3_effnet_f32_trace -- $EFF32
run effnet_f16_sq --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/prof_r3_effnet_f16_sq -- $EFF16
run effnet_f16_fetch --pmc FETCH_SIZE --output-format csv -d $OUT/prof_r3_effnet_f16_fetch -- $EFF16
run effnet_f16_write --pmc WRITE_SIZE --output-format csv -d $OUT/prof_r3_effnet_f16_write -- $EFF16
run resize_trace --kernel-trace --stats --output-format csv -d $OUT/prof_r3_resize_trace -- $RSZ
run resize_fetch --pmc FETCH_SIZE --output-format csv -d $OUT/prof_r3_resize_fetch -- $RSZ
run resize_write --pmc WRITE_SIZE --output-format csv -d $OUT/prof_r3_resize_write -- $RSZ
cd $R
for d in effnet_f16_trace effnet_f32_trace effnet_f16_sq effnet_f16_fetch effnet_f16_write resize_trace resize_fetch resize_write; do
  python tools/summarize_rocprof.py $OUT/prof_r3_$d $OUT/r3_$d.md "$d" || true
  find $OUT/prof_r3_$d -name '*.csv' -size +1M -delete
done
bash tools/profile_bench.sh r3
ls $OUT | grep r3_ | head -40
