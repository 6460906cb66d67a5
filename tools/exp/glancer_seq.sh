# Launch-ordered list of one serial 512-frame chunk of the glancer forward (kernel, grid, us) from a rocprofv3 kernel trace.  -> gpurun_out/r6_glancer_seq.txt
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gseq
timeout 300 env ${ADAF_LIB:+ADAF_LIB=$ADAF_LIB} rocprofv3 --kernel-trace --output-format csv -d /tmp/gseq -- python $R/tools/glancer_probe.py 1024 5 > /tmp/gseq.log 2>&1
f=$(find /tmp/gseq -name '*kernel_trace.csv' | head -1)
python - "$f" > $OUT/r6_glancer_seq.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ks = [r for r in rows if not r['Kernel_Name'].startswith(('void at::', '__amd_rocclr', 'pack_', 'fold_bn'))]
# the last forward: split at the stem kernel
idx = [i for i, r in enumerate(ks) if 'mb_stem' in r['Kernel_Name'] or 'stem' in r['Kernel_Name'].split('(')[0]]
last = ks[idx[-1]:]
tot = 0
for r in last:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    nm = re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', '')).replace('void ', '')
    print("%-78s grid %8d wg %4d  %8.1f us" % (nm[:78], int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Workgroup_Size_X']), d))
print("sum %.1f us over %d launches" % (tot, len(last)))
PY
cat $OUT/r6_glancer_seq.txt; tail -2 /tmp/gseq.log
