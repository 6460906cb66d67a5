#!/usr/bin/env python3
"""Per-launch timeline of ONE step out of a rocprofv3 kernel trace of tools/lat_trace_probe.py: the median over the last 40 steps of every
launch's duration and of the idle gap in front of it.  usage: lat_trace_report.py <rocprof dir> <out.md>"""
import csv
import glob
import os
import statistics
import sys

d, out = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "")) for r in csv.DictReader(open(f))))
# a step starts at every crop / stem launch that follows a gru_scan kernel
starts = [i for i, r in enumerate(rows) if ("crop_kernel" in r[2] or "stem7x7" in r[2]) and i > 0 and "gru_scan" in rows[i - 1][2]]
steps = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
steps = [s for s in steps if len(s) == len(steps[-1])][-40:]
n = len(steps[0])
lines = ["# one hot-path step at B = %s, T = 8, P = 96: per launch, median over %d steps (us)" % (os.environ.get("B", "1"), len(steps)), "",
         "| # | kernel | duration | gap before |", "|---|---|---|---|"]
tot_d = tot_g = 0.0
for i in range(n):
    dur = statistics.median((s[i][1] - s[i][0]) / 1e3 for s in steps)
    gap = statistics.median(((s[i][0] - s[i - 1][1]) / 1e3 if i else 0.0) for s in steps)
    tot_d += dur
    tot_g += gap
    lines.append("| %d | `%s` | %.2f | %.2f |" % (i, steps[0][i][2][:90], dur, gap))
span = statistics.median((s[-1][1] - s[0][0]) / 1e3 for s in steps)
lines += ["", "sum of durations %.1f us, sum of gaps %.1f us, first start -> last end %.1f us, %d launches" % (tot_d, tot_g, span, n)]
open(out, "w").write("\n".join(lines) + "\n")
print(lines[-1])
