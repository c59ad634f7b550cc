#!/usr/bin/env python3
"""GPU busy time of a rocprofv3 --kernel-trace CSV: union of the kernels' [start, end] intervals over the steady part of a run, the idle
share, the largest idle gaps and what ran on either side of them.  usage: trace_gaps.py <dir with *kernel_trace.csv> [skip_fraction]"""
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "?")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + (t1 - t0) * skip
rows = [r for r in rows if r[0] >= lo]
span = max(r[1] for r in rows) - rows[0][0]
busy, cur_s, cur_e, gaps, last = 0, rows[0][0], rows[0][1], [], rows[0]
per_queue = {}
for r in rows[1:]:
    if r[0] > cur_e:
        busy += cur_e - cur_s
        gaps.append((r[0] - cur_e, last[2], r[2]))
        cur_s, cur_e = r[0], r[1]
    else:
        cur_e = max(cur_e, r[1])
    if r[1] >= cur_e:
        last = r
busy += cur_e - cur_s
for r in rows:
    per_queue[r[3]] = per_queue.get(r[3], 0) + r[1] - r[0]
print("steady part: %.1f ms, device busy (union over queues) %.1f ms = %.1f %%, %d idle gaps" % (span / 1e6, busy / 1e6, 100 * busy / span, len(gaps)))
print("kernel time per queue (ms):", {k: round(v / 1e6, 1) for k, v in per_queue.items()})
hist = {}
for g, a, b in gaps:
    k = "%s -> %s" % (a[:38], b[:38])
    e = hist.setdefault(k, [0, 0])
    e[0] += g
    e[1] += 1
print("idle time by (kernel before -> kernel after):")
for k, (g, n) in sorted(hist.items(), key=lambda kv: -kv[1][0])[:18]:
    print("  %8.2f ms  %5d x %7.1f us  %s" % (g / 1e6, n, g / n / 1e3, k))
