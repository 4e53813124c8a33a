#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel, duration and the gap to the previous dispatch of the same
kernel (start - previous end) -- what a back-to-back launch loop pays between kernels."""
import csv, sys, collections
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"][:60]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for k, v in by.items():
    v.sort()
    if len(v) < 20:
        continue
    d = np.array([e - s for s, e in v]) / 1e3
    g = np.array([v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]) / 1e3
    g = g[g < 50]  # (drop the pauses between timing loops)
    print(f"{k:60s} n={len(v):5d} dur us median {np.median(d):7.2f} min {d.min():7.2f}   gap us median {np.median(g):6.2f} min {g.min():6.2f} p90 {np.percentile(g, 90):6.2f}")
