#!/bin/bash
# Round-4 rocprofv3 evidence (run on the GPU box through gpurun; summaries land in gpurun_out/r4_*.md, then
# `python tools/publish_profiles.py r4` copies them into profiles/ and derives profiles/r4_traffic.json / r4_effnet_traffic.json):
#   fullfwd: kernel traces of the full forward from uint8 clips, serial and on the model's two streams, + the overlap analysis
#   glancer: kernel trace + FETCH_SIZE + WRITE_SIZE of the MobileNetV2 glancer (1024 frames of 224^2)
#   effnet:  kernel trace + SQ + FETCH_SIZE + WRITE_SIZE of EfficientNet-B3 (1024 x 144^2, fp16 storage; whole-block kernels on)
#   bench:   tools/profile_bench.sh r4 (headline step: serial / 3-stream traces, MFMA / FETCH_SIZE / WRITE_SIZE passes)
# PMC passes run on their own (kernel-trace / stats domains only), one counter group per run.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; rocprofv3 "$@" > $OUT/r4_$name.log 2>&1; }
EFF16="python $R/tools/effnet_probe.py 1024 144 5 f16"
GL="python $R/tools/glancer_probe.py 1024"
run fullfwd_serial --kernel-trace --output-format csv -d $OUT/prof_r4_fullfwd_serial -- python $R/tools/fullfwd_probe.py serial 10
run fullfwd_two --kernel-trace --output-format csv -d $OUT/prof_r4_fullfwd_two -- python $R/tools/fullfwd_probe.py two 10
run glancer_trace --kernel-trace --stats --output-format csv -d $OUT/prof_r4_glancer_trace -- $GL
run glancer_fetch --pmc FETCH_SIZE --output-format csv -d $OUT/prof_r4_glancer_fetch -- $GL
run glancer_write --pmc WRITE_SIZE --output-format csv -d $OUT/prof_r4_glancer_write -- $GL
run effnet_f16_trace --kernel-trace --stats --output-format csv -d $OUT/prof_r4_effnet_f16_trace -- $EFF16
run effnet_f16_sq --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/prof_r4_effnet_f16_sq -- $EFF16
run effnet_f16_fetch --pmc FETCH_SIZE --output-format csv -d $OUT/prof_r4_effnet_f16_fetch -- $EFF16
run effnet_f16_write --pmc WRITE_SIZE --output-format csv -d $OUT/prof_r4_effnet_f16_write -- $EFF16
cd $R
python tools/overlap_report.py $OUT/prof_r4_fullfwd_serial $OUT/r4_fullfwd_trace_serial.md "serial" || true
python tools/overlap_report.py $OUT/prof_r4_fullfwd_two $OUT/r4_fullfwd_trace_2streams.md "two streams" || true
for d in glancer_trace glancer_fetch glancer_write effnet_f16_trace effnet_f16_sq effnet_f16_fetch effnet_f16_write; do
  python tools/summarize_rocprof.py $OUT/prof_r4_$d $OUT/r4_$d.md "$d" || true
done
grep -h "full forward" $OUT/r4_fullfwd_serial.log $OUT/r4_fullfwd_two.log > $OUT/r4_fullfwd_times.txt
for d in fullfwd_serial fullfwd_two glancer_trace glancer_fetch glancer_write effnet_f16_trace effnet_f16_sq effnet_f16_fetch effnet_f16_write; do
  find $OUT/prof_r4_$d -name '*.csv' -size +1M -delete
done
bash tools/profile_bench.sh r4
ls $OUT | grep r4_ | head -60
