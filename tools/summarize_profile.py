#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs that tools/profile_r.sh leaves under gpurun_out/prof/ into the small
tracked summaries under profiles/ (per-kernel stats of the default bench.py command and of the
roofline-only loop; per-launch HBM traffic of the gather-aggregate kernel from the PMC passes,
corrected as MI355X_MICROARCH.md's HBM section prescribes)."""
import collections
import csv
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
src = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "gpurun_out" / "prof"
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
dst = ROOT / "profiles"
dst.mkdir(exist_ok=True)


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    return name if len(name) < 90 else name[:87] + "..."


def kernel_stats(path: Path, out: Path, title: str) -> None:
    rows = list(csv.DictReader(open(path)))
    with open(out, "w") as f:
        f.write(f"# {title}\n# source: rocprofv3 --kernel-trace --stats ({path.name}); durations in microseconds\n")
        f.write("kernel,calls,total_us,avg_us,min_us,max_us,percent\n")
        for r in rows:
            f.write(f"\"{short(r['Name'])}\",{r['Calls']},{float(r['TotalDurationNs']) / 1e3:.1f},"
                    f"{float(r['AverageNs']) / 1e3:.2f},{float(r['MinNs']) / 1e3:.2f},{float(r['MaxNs']) / 1e3:.2f},"
                    f"{float(r['Percentage']):.2f}\n")


for wdir in sorted(src.glob("bench_c*")):
    if not wdir.is_dir():
        continue
    w = wdir.name.split("_", 1)[1]
    stats = list(wdir.rglob("bench_kernel_stats.csv"))
    if stats:
        kernel_stats(stats[0], dst / f"{tag}_{w}_bench_kernel_stats.csv", f"python3 bench.py --workload {w} --steps 100  (N=1)")
    lines = [l for l in (src / f"bench_{w}.log").read_text().splitlines() if l.startswith("{")]
    if lines:
        (dst / f"{tag}_{w}_bench.json").write_text(json.dumps(json.loads(lines[-1]), indent=1) + "\n")
if not (src / "roofline.log").exists():
    sys.exit(0)
kernel_stats(next((src / "roofline").rglob("roofline_kernel_stats.csv")), dst / f"{tag}_roofline_only_kernel_stats.csv",
             "python3 bench.py --roofline-only  (HBM-regime gather-aggregate loop + its copy calibration + the fused stack loop)")

def collect(pattern):
    pmc = {}
    for sub, fname in (("pmc_fetch", "fetch"), ("pmc_write", "write"), ("pmc_l2", "l2"), ("pmc_inst", "inst"),
                       ("pmc_busy", "busy"), ("pmc_wait", "wait")):
        found = list((src / sub).rglob(f"{fname}_counter_collection.csv")) if (src / sub).exists() else []
        if not found:
            continue
        p = found[0]
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(p)):
            if pattern in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            pmc[k] = {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
    return pmc


pmc = collect("k_aggregate_ring<0,")   # MODE 0 = GCN (the copy calibration is MODE 6)

log = (src / "roofline.log").read_text().strip().splitlines()
meas = json.loads([l for l in log if l.startswith("{")][-1])
summary = {"command": "rocprofv3 --pmc <counter> -- python3 bench.py --roofline-only (one pass per counter)",
           "kernel": "gnnb::k_aggregate_ring<GCN, float4, nt stores, one ring per workgroup>, width 128, BASELINE config 2 batch",
           "raw_counters_per_launch": pmc,
           "algorithmic_bytes_per_launch": meas["algorithmic_bytes_per_launch"],
           "events_us_per_launch": meas["us"]}
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    fetch = pmc["FETCH_SIZE"]["mean"] * 1024.0 * 2.0   # KiB units; gfx950 reports 1/2 of wide coalesced reads
    write = pmc["WRITE_SIZE"]["mean"] * 1024.0
    summary["hbm_traffic_bytes_per_launch"] = {"read_corrected_x2": fetch, "write": write, "total": fetch + write,
                                               "over_algorithmic": (fetch + write) / meas["algorithmic_bytes_per_launch"]}
if "TCC_HIT_sum" in pmc:
    h, m = pmc["TCC_HIT_sum"]["mean"], pmc["TCC_MISS_sum"]["mean"]
    summary["l2_hit_rate"] = h / (h + m)
(dst / f"{tag}_aggregate_pmc.json").write_text(json.dumps(summary, indent=2) + "\n")
print(json.dumps(summary, indent=2))

# ---- the fused 2-layer GCN stack kernel (dominant kernel of workload c2)
fs = meas.get("fused_stack")
g2 = collect("k_gcn2_fused")
if fs and g2:
    s2 = {"command": "rocprofv3 --pmc <counters> -- python3 bench.py --roofline-only (one pass per counter group)",
          "kernel": "gnnb::k_gcn2_fused<relu, KQ0=1, KQ1=8>, BASELINE config 2 batch",
          "raw_counters_per_launch": g2, "algorithmic_flops_per_launch": fs["flops"],
          "algorithmic_hbm_bytes_per_launch": fs["alg_bytes"], "events_us_per_launch": fs["us"]}
    if "FETCH_SIZE" in g2 and "WRITE_SIZE" in g2:
        fetch = g2["FETCH_SIZE"]["mean"] * 1024.0 * 2.0
        write = g2["WRITE_SIZE"]["mean"] * 1024.0
        s2["hbm_traffic_bytes_per_launch"] = {"read_corrected_x2": fetch, "write": write, "total": fetch + write,
                                              "over_algorithmic": (fetch + write) / fs["alg_bytes"]}
    if "SQ_INSTS_VALU" in g2 and "SQ_INSTS_MFMA" in g2:
        s2["instruction_mix_per_launch"] = {"valu_non_mfma": g2["SQ_INSTS_VALU"]["mean"] - g2["SQ_INSTS_MFMA"]["mean"],
                                            "mfma": g2["SQ_INSTS_MFMA"]["mean"],
                                            "salu": g2.get("SQ_INSTS_SALU", {}).get("mean"),
                                            "lds": g2.get("SQ_INSTS_LDS", {}).get("mean")}
    (dst / f"{tag}_gcn2_pmc.json").write_text(json.dumps(s2, indent=2) + "\n")
    print(json.dumps(s2, indent=2))
