#!/usr/bin/env python3
"""Prints the last region (kernels separated by > 1 ms gaps) of a rocprofv3 kernel trace of tools/region_trace.py as a
timeline: start, duration, short kernel name.  usage: trace_timeline.py kernel_trace.csv [max_rows]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
regions, cur = [], []
for row in rows:
    if cur and row[0] - max(x[1] for x in cur) > 1_000_000:
        regions.append(cur)
        cur = []
    cur.append(row)
if cur:
    regions.append(cur)
reg = regions[-2] if len(regions) > 1 else regions[-1]
t0 = reg[0][0]
print("span %.1f us, %d kernels" % ((max(x[1] for x in reg) - t0) / 1e3, len(reg)))
for s, e, n, q in reg[: int(sys.argv[2]) if len(sys.argv) > 2 else 80]:
    name = n.split("(")[0].split("::")[-1].split("<")[0][:22]
    print("%8.1f %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, name))
