#!/bin/bash
# per-kernel table (us per forward) of the EfficientNet-B3 forward, fp16 storage, 1024 x 144^2: rocprofv3 kernel trace of tools/effnet_probe.py
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/efk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/efk -- python $R/tools/effnet_probe.py 1024 144 5 ${1:-f16} > /tmp/efk.log 2>&1
tail -1 /tmp/efk.log
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/efk/**/*kernel_stats.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in ("copyBuffer", "pack_", "fold_bn", "at::native", "transpose", "fillBuffer")): continue
    rows.append((float(r["TotalDurationNs"]) / 8e3, int(r["Calls"]) / 8, n.replace("(anonymous namespace)::", "")[:70]))
rows.sort(reverse=True)
for t, c, n in rows: print("%8.1f us/fwd  %4.1f calls  %s" % (t, c, n))
print("total %.1f" % sum(r[0] for r in rows))
PY
