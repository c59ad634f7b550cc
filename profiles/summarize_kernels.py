#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace --stats output directory: calls, total / average duration, share."""
import csv, glob, re, sys
files = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not files:
    sys.exit("no *kernel_stats.csv under " + sys.argv[1])
rows = list(csv.DictReader(open(files[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("%-86s %6s %11s %10s %6s" % ("kernel", "calls", "total ms", "avg us", "%"))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    name = re.sub(r"\(.*\)$", "", name)
    print("%-86s %6d %11.3f %10.1f %6.2f" % (name[:86], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6,
                                                float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("total kernel time %.3f ms" % (tot / 1e6))
