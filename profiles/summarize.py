#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace stats + PMC passes) into a small text summary.
usage: python profiles/summarize.py gpurun_out/<tag>"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


def short(name):
    return name.split("(")[0].replace("void ", "")[:60]


print("== kernel trace (per-kernel durations from kernel_trace.csv)")
for f in find("*kernel_trace.csv"):
    durs = defaultdict(list)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            durs[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in durs.values())
    print("%-62s %8s %12s %12s %12s %7s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct"))
    for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
        print("%-62s %8d %12.2f %12.2f %12.2f %6.1f%%" % (k, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3, 100.0 * sum(v) / tot))
for f in find("*kernel_stats.csv"):
    print("== %s" % os.path.relpath(f, root))
    print(open(f).read()[:3000])

print("== PMC (per-kernel mean counter value per dispatch)")
for f in find("*counter_collection.csv"):
    vals = defaultdict(list)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            vals[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(vals.items()):
        print("%-62s %-12s dispatches %5d  mean %.1f" % (k, c, len(v), sum(v) / len(v)))
