set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_r6_fullfwd_serial -- python $R/tools/fullfwd_probe.py serial 10 > $OUT/r6_fullfwd_serial.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_r6_fullfwd_two -- python $R/tools/fullfwd_probe.py two 10 > $OUT/r6_fullfwd_two.log 2>&1
cd $R
python tools/overlap_report.py $OUT/prof_r6_fullfwd_serial $OUT/r6_fullfwd_trace_serial.md "serial" || true
python tools/overlap_report.py $OUT/prof_r6_fullfwd_two $OUT/r6_fullfwd_trace_2streams.md "two streams" || true
grep -h "full forward" $OUT/r6_fullfwd_serial.log $OUT/r6_fullfwd_two.log
find $OUT/prof_r6_fullfwd_serial $OUT/prof_r6_fullfwd_two -name '*.csv' -size +1M -delete
head -30 $OUT/r6_fullfwd_trace_serial.md | cut -c1-160
