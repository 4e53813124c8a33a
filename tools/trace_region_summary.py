#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace of tools/region_trace.py: per region (kernels separated by > 1 ms gaps) the span, the
busy time of k_gcn2_zf, and the idle gaps between consecutive zf kernels."""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
regions, cur = [], []
for s, e, n in rows:
    if cur and s - max(x[1] for x in cur) > 1_000_000:
        regions.append(cur)
        cur = []
    cur.append((s, e, n))
if cur:
    regions.append(cur)
for reg in regions[-4:]:
    t0 = reg[0][0]
    span = (max(x[1] for x in reg) - t0) / 1e3
    zf = [(s, e) for s, e, n in reg if "gcn2_zf" in n]
    if not zf:
        continue
    print("region: %d kernels, span %.1f us; first kernel %s; zf kernels %d: first starts at %.1f, last ends at %.1f" % (
        len(reg), span, reg[0][2][:24], len(zf), (zf[0][0] - t0) / 1e3, (zf[-1][1] - t0) / 1e3))
    print("  zf start / dur (us):", " ".join("%.0f/%.0f" % ((s - t0) / 1e3, (e - s) / 1e3) for s, e in zf))
    others = [(s, e, n) for s, e, n in reg if "gcn2_zf" not in n]
    print("  last 4 other kernels end at:", " ".join("%s@%.0f-%.0f" % (n.split("::")[-1][:12], (s - t0) / 1e3, (e - t0) / 1e3) for s, e, n in others[-4:]))
