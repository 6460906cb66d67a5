#!/usr/bin/env python3
"""Do kernels of different streams co-run, and does co-running conserve work?  Reads a rocprofv3 --kernel-trace CSV (start / end time
stamps, queue id per dispatch) and writes: wall time of the traced region, the sum of the kernel durations, the share of the wall time
with 0 / 1 / >= 2 kernels in flight, and per kernel family the dispatch count and mean duration.  Comparing a serial run with a
two-stream run of the same work: if the kernels co-run (>= 2 in flight most of the time) but the wall time does not shrink, their
durations stretch by the same factor -- they share the resource that bounds them.
usage: overlap_report.py <rocprof dir> <out.md> "<title>" [skip_first_fraction=0.35]"""
import collections
import csv
import glob
import os
import re
import sys


def family(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"<.*", "", name)
    m = re.match(r"_ZN\d+_GLOBAL__N_\d+(\d\d)([A-Za-z_0-9]+)", name)
    if m:
        name = m.group(2)[:int(m.group(1))]
    return name.split("(")[0][:60]


def main():
    d, out, title = sys.argv[1], sys.argv[2], sys.argv[3]
    skip = float(sys.argv[4]) if len(sys.argv) > 4 else 0.35
    rows = []
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
    rows.sort()
    # steady state only: drop the first `skip` of the dispatches (model load, warm-up, first passes)
    rows = rows[int(len(rows) * skip):]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    wall = (t1 - t0) / 1e3
    ev = []
    for s, e, _, _ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last, hist = 0, t0, collections.Counter()
    for ts, dlt in ev:
        hist[min(depth, 3)] += ts - last
        last, depth = ts, depth + dlt
    total = sum(hist.values()) or 1
    ksum = sum(e - s for s, e, _, _ in rows) / 1e3
    fam = collections.defaultdict(lambda: [0, 0.0])
    for s, e, n, _ in rows:
        a = fam[family(n)]
        a[0] += 1
        a[1] += (e - s) / 1e3
    queues = len(set(r[3] for r in rows))
    lines = ["# %s" % title, "",
             "steady-state window (last %.0f %% of the dispatches): %d dispatches on %d queue(s), wall %.1f us, sum of kernel durations %.1f us "
             "(= %.2f x wall)" % (100 * (1 - skip), len(rows), queues, wall, ksum, ksum / wall), "",
             "| kernels in flight | share of the wall time |", "|---|---|"]
    for k, label in ((0, "0 (gaps)"), (1, "1"), (2, "2"), (3, ">= 3")):
        lines.append("| %s | %.1f %% |" % (label, 100.0 * hist[k] / total))
    lines += ["", "| kernel family | dispatches | total us | mean us |", "|---|---|---|---|"]
    for k, (n, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:30]:
        lines.append("| `%s` | %d | %.1f | %.2f |" % (k, n, t, t / n))
    open(out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
