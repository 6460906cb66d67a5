#!/bin/bash
# Round-6 rocprofv3 evidence in one command (run on the GPU box through gpurun; summaries land in gpurun_out/r6_*.md, then
# `python tools/publish_profiles.py r6` copies them into profiles/ and derives profiles/r6_traffic.json / r6_glancer_traffic.json):
#   glancer: tools/profile_r6_glancer.sh r6 (kernel traces paired / serial, FETCH_SIZE, WRITE_SIZE, SQ passes of the 1024-frame glancer)
#   split:   the headline step with --math split_bf16: serial kernel trace, MFMA-busy + LDS PMC pass
#   bench:   tools/profile_bench.sh r6 (headline step on the fp32 pipe: serial / 3-stream traces, MFMA / FETCH_SIZE / WRITE_SIZE passes)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
bash $R/tools/profile_r6_glancer.sh r6
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; rocprofv3 "$@" > $OUT/r6_$name.log 2>&1; }
SPLIT="python $R/bench.py --steps 20 --warmup 4 --skip-extras --streams 1 --cpu-baseline 0 --math split_bf16"
SPLITPMC="python $R/bench.py --steps 4 --warmup 2 --skip-extras --profile-steps 1 --streams 1 --cpu-baseline 0 --math split_bf16"
run split_trace --kernel-trace --stats --output-format csv -d $OUT/prof_r6_split_trace -- $SPLIT
run split_mfma --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $OUT/prof_r6_split_mfma -- $SPLITPMC
run split_lds --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_INSTS_MFMA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_r6_split_lds -- $SPLITPMC
cd $R
for d in split_trace split_mfma split_lds; do
  python tools/summarize_rocprof.py $OUT/prof_r6_$d $OUT/r6_$d.md "$d" || true
  find $OUT/prof_r6_$d -name '*.csv' -size +1M -delete
done
bash tools/profile_bench.sh r6
ls $OUT | grep r6_ | head -60
