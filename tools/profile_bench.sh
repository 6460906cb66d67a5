#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline entry (run on the GPU box through gpurun):
#   1. kernel trace of the serial bench command (--streams 1), 2. kernel trace of the default 3-stream command,
#   3.-5. PMC passes (MFMA busy / FETCH_SIZE / WRITE_SIZE, each in its own run, kernel-trace/stats domains only).
# Summaries: python tools/summarize_rocprof.py gpurun_out/<dir> profiles/<name>.md "<command>"
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r2}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
SER="python $R/bench.py --steps 20 --warmup 4 --skip-extras --streams 1 --cpu-baseline 0"
PMC="python $R/bench.py --steps 4 --warmup 2 --skip-extras --profile-steps 1 --streams 1 --cpu-baseline 0"
PAR="python $R/bench.py --steps 30 --warmup 6 --skip-extras --cpu-baseline 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_trace1 -- $SER > $OUT/prof_${TAG}_trace1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_trace3 -- $PAR > $OUT/prof_${TAG}_trace3.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/prof_${TAG}_mfma -- $PMC > $OUT/prof_${TAG}_mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_fetch -- $PMC > $OUT/prof_${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_write -- $PMC > $OUT/prof_${TAG}_write.log 2>&1
cd $R
for d in trace1 trace3 mfma fetch write; do
  python tools/summarize_rocprof.py $OUT/prof_${TAG}_$d $OUT/${TAG}_$d.md "$d" || true
  # keep the merge-back small: the raw CSVs stay on the box
  find $OUT/prof_${TAG}_$d -name '*.csv' -size +2M -delete
done
ls -la $OUT | tail -20
