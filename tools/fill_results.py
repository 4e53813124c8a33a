#!/usr/bin/env python3
"""Regenerates the results tables of DESIGN.md and README.md (between the R6_TABLE markers) from profiles/r06_*_bench.json
(`bench.py --workload w --steps 100`, builder-run) and profiles/r06_driver_command.json (`bench.py --gpus 1 --steps 20
--warmup 5`: the driver's command, whose line carries c3 / c4 / c5 as `other_configs`)."""
import json
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
P = ROOT / "profiles"
NAMES = [("c2", "C2 GCN L2 d128 B4096"), ("c3", "C3 GIN L3 d128 B4096"), ("c4", "C4 PNA L3 d128 B8192 (both promises)"),
         ("c5", "C5 SAGE L2 d256 B8192/GPU")]


def fmt(v):
    return f"{v / 1e6:.2f} M" if v >= 1e6 else f"{v:,.0f}"


drv = json.loads((P / "r06_driver_command.json").read_text())
others = {o["name"]: o for o in drv.get("other_configs", [])}
rows = []
for w, name in NAMES:
    d = json.loads((P / f"r06_{w}_bench.json").read_text())
    r = d["roofline"]
    g = d["roofline_gather_aggregate"]
    if w == "c2":
        dv = f"**{fmt(drv['value'])}**, {drv['ms_per_step'] * 1e3:.1f} µs"
        rp = drv.get("opt_in_math_reduced_precision")
        if rp:
            dv += (f" (opt-in REDUCED precision, not `value`: bf16x3 {fmt(rp['bf16x3']['value'])}, {rp['bf16x3']['ms_per_step'] * 1e3:.1f} µs; "
                   f"f16x3 {fmt(rp['f16x3']['value'])}, {rp['f16x3']['ms_per_step'] * 1e3:.1f} µs)")
    else:
        o = others[w]
        dv = f"**{fmt(o['value'])}**, {o['ms_per_step'] * 1e3:.1f} µs"
        m, m3 = o.get("opt_in_math_bf16x6"), o.get("opt_in_math_f16x3_reduced_precision")
        if m3 and not m:
            dv += f" (opt-in, not `value`: f16x3, reduced precision, {fmt(m3['value'])}, {m3['ms_per_step'] * 1e3:.1f} µs)"
        if m:
            dv += f" (opt-in, not `value`: bf16x6 GEMMs {fmt(m['value'])}, {m['ms_per_step'] * 1e3:.1f} µs"
            if m3:
                dv += f"; f16x3, reduced precision, {fmt(m3['value'])}, {m3['ms_per_step'] * 1e3:.1f} µs"
            dv += ")"
    kern = re.sub(r" \(.*", "", r["kernel"])
    solo = f"`{kern}` {r['us_per_launch']:.1f} µs = {r['achieved']:.1f} TFLOP/s = **{r['frac']:.3f}**"
    ip = r.get("in_pipeline")
    if ip:
        solo += f"; in the pipeline {ip['frac']:.3f}"
    agg = f"{g['frac']:.2f} ({g['us_per_launch']:.1f} µs)"
    wk = g.get("workload_kind")
    if wk:
        agg += f" / {wk['frac']:.2f} ({wk['us_per_launch']:.1f} µs)"
    pa = g.get("pna_product_aggregate")
    if pa:
        agg += f"; `k_pna_pagg` {pa['frac']:.2f} ({pa['us_per_launch']:.1f} µs)"
    cpu = d["cpu_baseline"]
    rows.append(f"| {name} | {dv} | {fmt(d['value'])}, {d['ms_per_step'] * 1e3:.1f} µs | {solo} | {agg} | {fmt(cpu['value'])} |")
table = ("| workload | the driver's command (`--steps 20`): graphs/s, step | `--steps 100` | dominant kernel: solo launch time = achieved = fraction of the fp32 MFMA peak; the same flops over the timed step | gather-aggregate, fraction of 8 TB/s: GCN kind / the workload's own kind | reference C++ on one host core, graphs/s |\n"
         "|---|---|---|---|---|---|\n" + "\n".join(rows))
for f in ("DESIGN.md", "README.md"):
    p = ROOT / f
    s = p.read_text()
    s2 = re.sub(r"<!-- R6_TABLE_BEGIN -->.*?<!-- R6_TABLE_END -->", "<!-- R6_TABLE_BEGIN -->\n" + table + "\n<!-- R6_TABLE_END -->", s, flags=re.S)
    if s2 != s:
        p.write_text(s2)
        print("updated", f)
print(table)
