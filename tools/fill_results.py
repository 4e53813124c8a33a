#!/usr/bin/env python3
"""Regenerates the results tables of DESIGN.md and README.md (between the R4_TABLE markers) from profiles/r04_*_bench.json."""
import json, re
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
P = ROOT / "profiles"
NAMES = [("c2", "C2 GCN L2 d128 B4096", "`k_gcn2_zf`"), ("c3", "C3 GIN L3 d128 B4096", "`k_gcn2_fused<GIN>`"),
         ("c3t", "C3t = C3 with the data set's heavy tail", "`k_gcn2_fused<GIN>` (+ large segment)"),
         ("c4", "C4 PNA L3 d128 B8192 (max_degree promise)", "degree-class GEMM `k_linear_dma` (K = 5F)"), ("c5", "C5 SAGE L2 d256 B8192/GPU", "K=512 GEMM `k_linear_dma`"),
         ("ref6_gcn", "ref6 GCN (6 layers 128→64, MLP 4×64) B4096", "`k_gcn2_fused` (6 layers)"), ("ref6_gin", "ref6 GIN", "`k_gcn2_fused<GIN>` (6 layers)"),
         ("ref6_sage", "ref6 SAGE", "K=256 GEMM `k_linear_dma`"), ("ref6_pna", "ref6 PNA (max_degree promise)", "degree-class GEMM `k_linear_dma` (K = 5F)")]


def fmt(v):
    return f"{v / 1e6:.2f} M" if v >= 1e6 else f"{v:,.0f}"


rows = []
for w, name, kern in NAMES:
    d = json.loads((P / f"r04_{w}_bench.json").read_text())
    r, ga, cb = d["roofline"], d.get("roofline_gather_aggregate") or {}, d.get("cpu_baseline") or {}
    wk = ga.get("workload_kind") or {}
    unit = "TFLOP/s" if r["bound"] == "mfma" else "GB/s"
    agg = f"{ga.get('frac', 0):.2f} ({ga.get('us_per_launch', 0):.1f} µs)" if ga else ""
    if wk:
        agg += f" / {wk['frac']:.2f} ({wk['us_per_launch']:.1f} µs)"
    rows.append(f"| {name} | **{fmt(d['value'])}** | {d['ms_per_step'] * 1000:.1f} µs | {kern} {r['us_per_launch']:.1f} µs = {r['achieved']:.1f} {unit} = **{r['frac']:.3f}** | {agg} | {cb.get('value', 0):,.0f} |")
table = ("| workload | graphs/s | step | dominant kernel (roofline: launch time = achieved = fraction of the fp32 MFMA peak) | gather-aggregate, fraction of 8 TB/s: GCN kind / the workload's own kind | reference C++ on one host core, graphs/s |\n"
         "|---|---|---|---|---|---|\n" + "\n".join(rows))
for f in ("DESIGN.md", "README.md"):
    p = ROOT / f
    s = p.read_text()
    s2 = re.sub(r"<!-- R4_TABLE_BEGIN -->.*?<!-- R4_TABLE_END -->", "<!-- R4_TABLE_BEGIN -->\n" + table + "\n<!-- R4_TABLE_END -->", s, flags=re.S)
    if s2 != s:
        p.write_text(s2)
        print("updated", f)
