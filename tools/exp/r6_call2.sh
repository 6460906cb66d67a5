set -u
mkdir -p gpurun_out
timeout 600 python tools/exp/strip_ab.py 1024 4 2>&1 | tail -20
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r6s_trace_serial -- python $GRAFT_REPO_ROOT/tools/glancer_probe.py 1024 5 > $GRAFT_REPO_ROOT/gpurun_out/r6s_trace_serial.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_rocprof.py gpurun_out/prof_r6s_trace_serial gpurun_out/r6s_trace_serial.md "serial" || true
find gpurun_out/prof_r6s_trace_serial -name '*.csv' -size +1M -delete
head -14 gpurun_out/r6s_trace_serial.md | cut -c1-160
