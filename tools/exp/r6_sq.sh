set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_r6s_sq -- python $R/tools/glancer_probe.py 1024 5 > $OUT/r6s_sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_MFMA --output-format csv -d $OUT/prof_r6s_sq2 -- python $R/tools/glancer_probe.py 1024 5 > $OUT/r6s_sq2.log 2>&1
cd $R
python tools/summarize_rocprof.py $OUT/prof_r6s_sq $OUT/r6s_sq.md sq || true
python tools/summarize_rocprof.py $OUT/prof_r6s_sq2 $OUT/r6s_sq2.md sq2 || true
find $OUT/prof_r6s_sq $OUT/prof_r6s_sq2 -name '*.csv' -size +1M -delete
grep -E "mb_(stem|block|expand)" $OUT/r6s_sq.md | cut -c1-200
grep -E "mb_(stem|block|expand)" $OUT/r6s_sq2.md | cut -c1-200
tail -3 $OUT/r6s_sq2.log
