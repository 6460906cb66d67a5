set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for k in ${ABLS:-0 1 2 3 4 12 7 15 31}; do
  if [ $k = 0 ]; then unset ADAF_LIB; else export ADAF_LIB=$R/adafocus_amd/csrc/exp_build/libadafocus_hip_abl$k.so; fi
  rm -rf /tmp/abl_$k
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_$k -- python $R/tools/glancer_probe.py 1024 5 > /tmp/abl_$k.log 2>&1
  f=$(find /tmp/abl_$k -name '*kernel_stats.csv' | head -1)
  echo "== MBS_ABL=$k  $(grep 'glancer (from' /tmp/abl_$k.log)"
  python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'mb_' in n: print("   %-48s calls %s avg %.1f us" % (n[:48], r['Calls'], float(r['AverageNs'])/1e3))
PY
done > $OUT/r6_abl.txt 2>&1
cat $OUT/r6_abl.txt
