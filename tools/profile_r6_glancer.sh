#!/bin/bash
# Round-6 rocprofv3 evidence for the glancer (rows a10 / f2; run on the GPU box through gpurun, summaries land in gpurun_out/<tag>_*.md, then
# `python tools/publish_profiles.py <tag>` copies them into profiles/ and derives profiles/<tag>_glancer_traffic.json):
#   glancer_trace        kernel trace of the default plan (two frame chunks side by side on two streams)
#   glancer_trace_serial the same forward with chunk pairing off (adaf_mobilenetv2_set_fusion bit 2): every kernel alone on the device
#   glancer_fetch / glancer_write / glancer_sq   PMC passes, one counter group per run (kernel-trace / stats domains only)
# usage: profile_r6_glancer.sh [tag=r6]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r6}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; rocprofv3 "$@" > $OUT/${TAG}_$name.log 2>&1; }
GL="python $R/tools/glancer_probe.py 1024"
run glancer_trace --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_glancer_trace -- $GL
run glancer_trace_serial --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_glancer_trace_serial -- $GL 5
run glancer_fetch --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_glancer_fetch -- $GL
run glancer_write --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_glancer_write -- $GL
run glancer_sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_${TAG}_glancer_sq -- $GL 5
cd $R
for d in glancer_trace glancer_trace_serial glancer_fetch glancer_write glancer_sq; do
  python tools/summarize_rocprof.py $OUT/prof_${TAG}_$d $OUT/${TAG}_$d.md "$d" || true
  find $OUT/prof_${TAG}_$d -name '*.csv' -size +1M -delete
done
grep -h "glancer (from" $OUT/${TAG}_glancer_*.log
