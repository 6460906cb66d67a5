#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace / counter collection) into the small summaries
committed under profiles/.  Usage: summarize_rocprof.py <dir> <out.md> [title]"""
import collections
import csv
import glob
import os
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name if len(name) < 110 else name[:107] + "..."


def main():
    d, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else d
    lines = ["# %s" % title, ""]
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        agg = collections.defaultdict(lambda: [0, 0.0, []])
        for r in csv.DictReader(open(f)):
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += dur
            a[2].append(dur)
        tot = sum(v[1] for v in agg.values())
        lines += ["## kernel trace: %s" % os.path.basename(f), "",
                  "| kernel | calls | total us | avg us | median us | % |", "|---|---|---|---|---|---|"]
        for k, (n, t, ds) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
            ds.sort()
            lines.append("| `%s` | %d | %.1f | %.2f | %.2f | %.1f |" % (k, n, t, t / n, ds[len(ds) // 2], 100 * t / tot))
        lines.append("")
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
        for r in csv.DictReader(open(f)):
            a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        lines += ["## counters: %s" % os.path.relpath(f, d), "", "| kernel | counter | dispatches | sum | per dispatch |", "|---|---|---|---|---|"]
        for k, cs in agg.items():
            for c, (n, v) in cs.items():
                lines.append("| `%s` | %s | %d | %.4g | %.4g |" % (k, c, n, v, v / n))
        lines.append("")
    open(out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
