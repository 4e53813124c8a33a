#!/usr/bin/env python3
"""Print a steady-state window of a rocprofv3 --kernel-trace CSV as a timeline (start / end in us relative to the window,
queue, kernel): how the three batches in flight interleave on the chip.  usage: timeline.py trace.csv [first] [count]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
count = int(sys.argv[3]) if len(sys.argv) > 3 else 45
win = rows[first:first + count]
t0 = int(win[0]["Start_Timestamp"])
for r in win:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].replace("gnnb::", "").split("(")[0][:28]
    print(f"{s:9.2f} {e:9.2f} {e - s:7.2f}  q{r.get('Queue_Id', '?'):>3s}  {name}")
