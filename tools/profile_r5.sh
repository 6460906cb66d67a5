#!/bin/bash
# Round-5 rocprofv3 evidence (run on the GPU box through gpurun; summaries land in gpurun_out/r5_*.md, then `python tools/publish_profiles.py r5`
# copies them into profiles/ and derives profiles/r5_traffic.json):
#   bench:  tools/profile_bench.sh r5 (headline step on the fp32 pipe: serial / 3-stream traces, MFMA + LDS / FETCH_SIZE / WRITE_SIZE passes)
#   split:  the same step with --math split_bf16 (also.split_bf16): serial kernel trace, MFMA-busy + LDS-bank-conflict PMC pass
# PMC passes run on their own (kernel-trace / stats domains only), one counter group per run.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; rocprofv3 "$@" > $OUT/r5_$name.log 2>&1; }
SPLIT="python $R/bench.py --steps 20 --warmup 4 --skip-extras --streams 1 --cpu-baseline 0 --math split_bf16"
SPLITPMC="python $R/bench.py --steps 4 --warmup 2 --skip-extras --profile-steps 1 --streams 1 --cpu-baseline 0 --math split_bf16"
run split_trace --kernel-trace --stats --output-format csv -d $OUT/prof_r5_split_trace -- $SPLIT
run split_mfma --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $OUT/prof_r5_split_mfma -- $SPLITPMC
cd $R
for d in split_trace split_mfma; do
  python tools/summarize_rocprof.py $OUT/prof_r5_$d $OUT/r5_$d.md "$d" || true
  find $OUT/prof_r5_$d -name '*.csv' -size +1M -delete
done
bash tools/profile_bench.sh r5
ls $OUT | grep r5_ | head -40
