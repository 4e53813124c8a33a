#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs that tools/profile_r.sh leaves under gpurun_out/prof/ into the small tracked summaries under
profiles/ (usage: summarize_profile.py [gpurun_out/prof] [tag]).  Per workload w:
  <tag>_<w>_bench_kernel_stats.csv     per-kernel stats of `bench.py --workload w --steps 100` under rocprofv3
  <tag>_<w>_bench.json                 the JSON line of the same command run WITHOUT the profiler (profile_r.sh says why)
  <tag>_<w>_roofline_kernel_stats.csv  ... of `bench.py --workload w --roofline-only`
  <tag>_<w>_aggregate_pmc.json         HBM traffic per launch of the GCN gather-aggregate kernel at this workload's width
                                       (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, as MI355X_MICROARCH.md's HBM section prescribes)
  <tag>_<w>_gcn2_pmc.json              the same + instruction mix / wait classes of the conv-stack kernel, where there is one"""
import collections
import csv
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
src = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "gpurun_out" / "prof"
tag = sys.argv[2] if len(sys.argv) > 2 else "r05"
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


def collect(w, pattern):
    pmc = {}
    for sub in sorted(src.glob(f"pmc_{w}_*")):
        if not sub.is_dir():
            continue
        for p in sub.rglob("*counter_collection.csv"):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(p)):
                if pattern in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in acc.items():
                pmc[k] = {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
    return pmc


def traffic(pmc, alg):
    if "FETCH_SIZE" not in pmc or "WRITE_SIZE" not in pmc:
        return None
    fetch = pmc["FETCH_SIZE"]["mean"] * 1024.0 * 2.0   # KiB units; gfx950 reports 1/2 of wide coalesced reads
    write = pmc["WRITE_SIZE"]["mean"] * 1024.0
    return {"read_corrected_x2": fetch, "write": write, "total": fetch + write, "over_algorithmic": (fetch + write) / alg}


for wdir in sorted(src.glob("bench_*")):
    if not wdir.is_dir():
        continue
    w = wdir.name.split("_", 1)[1]
    stats = list(wdir.rglob("bench_kernel_stats.csv"))
    if stats:
        kernel_stats(stats[0], dst / f"{tag}_{w}_bench_kernel_stats.csv", f"python3 bench.py --workload {w} --steps 100  (N=1)")
    plain = src / f"plain_{w}.log"
    lines = [l for l in plain.read_text().splitlines() if l.startswith("{")] if plain.exists() else []
    if lines:
        (dst / f"{tag}_{w}_bench.json").write_text(json.dumps(json.loads(lines[-1]), indent=1) + "\n")
    rl = src / f"roofline_{w}.log"
    if not rl.exists():
        print(w, "done (line + kernel table)")
        continue
    rstats = list((src / f"roofline_{w}").rglob("roofline_kernel_stats.csv"))
    if rstats:
        kernel_stats(rstats[0], dst / f"{tag}_{w}_roofline_kernel_stats.csv",
                     f"python3 bench.py --workload {w} --roofline-only  (HBM-regime gather-aggregate loop + its copy calibration + the conv-stack loop)")
    meas = [l for l in rl.read_text().splitlines() if l.startswith("{")]
    if not meas:
        continue
    meas = json.loads(meas[-1])
    pmc = collect(w, "k_aggregate_ring<0,")   # MODE 0 = GCN (the copy calibration is MODE 6)
    if pmc:
        summary = {"command": f"rocprofv3 --pmc <counter> -- python3 bench.py --workload {w} --roofline-only (one pass per counter group)",
                   "workload": w, "kernel": "gnnb::k_aggregate_ring<GCN, float4, nt stores>, this workload's batch and width",
                   "raw_counters_per_launch": pmc, "algorithmic_bytes_per_launch": meas["algorithmic_bytes_per_launch"],
                   "events_us_per_launch": meas["us"]}
        t = traffic(pmc, meas["algorithmic_bytes_per_launch"])
        if t:
            summary["hbm_traffic_bytes_per_launch"] = t
        if "TCC_HIT_sum" in pmc:
            h, m = pmc["TCC_HIT_sum"]["mean"], pmc["TCC_MISS_sum"]["mean"]
            summary["l2_hit_rate"] = h / (h + m)
        (dst / f"{tag}_{w}_aggregate_pmc.json").write_text(json.dumps(summary, indent=2) + "\n")
    # the aggregate kind this workload's own layers run (SUM / MEAN / PNA with four output matrices)
    own = meas.get("workload_kind")
    if own:
        mode = {"sum": 1, "mean": 2, "pna": 3}[own["kind"]]
        pk = collect(w, f"k_aggregate_ring<{mode},")
        if pk:
            so = {"command": f"rocprofv3 --pmc <counter> -- python3 bench.py --workload {w} --roofline-only (one pass per counter group)",
                  "workload": w, "kernel": f"gnnb::k_aggregate_ring<{own['kind'].upper()}, float4, nt stores>, this workload's batch and width",
                  "raw_counters_per_launch": pk, "algorithmic_bytes_per_launch": own["algorithmic_bytes_per_launch"],
                  "events_us_per_launch": own["us"]}
            t = traffic(pk, own["algorithmic_bytes_per_launch"])
            if t:
                so["hbm_traffic_bytes_per_launch"] = t
            (dst / f"{tag}_{w}_aggregate_{own['kind']}_pmc.json").write_text(json.dumps(so, indent=2) + "\n")
    fs = meas.get("fused_stack")
    g2 = collect(w, "k_gcn2_")
    if fs and g2:
        s2 = {"command": f"rocprofv3 --pmc <counters> -- python3 bench.py --workload {w} --roofline-only (one pass per counter group)",
              "workload": w, "kernel": f"the conv-stack kernel of this workload ({meas.get('stack_path')})",
              "raw_counters_per_launch": g2, "algorithmic_flops_per_launch": fs["flops"],
              "algorithmic_hbm_bytes_per_launch": fs["alg_bytes"], "events_us_per_launch": fs["us"]}
        t = traffic(g2, fs["alg_bytes"])
        if t:
            s2["hbm_traffic_bytes_per_launch"] = t
        if "SQ_INSTS_VALU" in g2 and "SQ_INSTS_MFMA" in g2:
            s2["instruction_mix_per_launch"] = {"valu_non_mfma": g2["SQ_INSTS_VALU"]["mean"] - g2["SQ_INSTS_MFMA"]["mean"],
                                                "mfma": g2["SQ_INSTS_MFMA"]["mean"],
                                                "salu": g2.get("SQ_INSTS_SALU", {}).get("mean"),
                                                "lds": g2.get("SQ_INSTS_LDS", {}).get("mean")}
        if "SQ_LDS_BANK_CONFLICT" in g2 and "SQ_LDS_IDX_ACTIVE" in g2:
            s2["lds_bank_conflict_share_of_lds_cycles"] = g2["SQ_LDS_BANK_CONFLICT"]["mean"] / g2["SQ_LDS_IDX_ACTIVE"]["mean"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in g2 and "SQ_BUSY_CYCLES" in g2:
            s2["note_busy"] = "SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs: / 1024 = matrix-pipe cycles per SIMD"
        (dst / f"{tag}_{w}_gcn2_pmc.json").write_text(json.dumps(s2, indent=2) + "\n")
    print(w, "done")

# the swizzled build of k_gcn2_zf (profile_r.sh): conflicts and time beside the shipped kernel's
swz = src / "swz_roofline.log"
if swz.exists():
    meas = [l for l in swz.read_text().splitlines() if l.startswith("{")]
    pm = {}
    for sub in ("pmc_swz_lds", "pmc_swz_inst"):
        for p in (src / sub).rglob("*counter_collection.csv"):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(p)):
                if "k_gcn2_zf" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in acc.items():
                pm[k] = {"launches": len(v), "mean": sum(v) / len(v)}
    if meas and pm:
        m = json.loads(meas[-1])
        out = {"what": "k_gcn2_zf built with -DZF_SWZ=1 (H / Z rows unpadded, 16-B chunks XOR-swizzled by the row index): NOT the shipped kernel",
               "events_us_per_launch": (m.get("fused_stack") or {}).get("us"), "raw_counters_per_launch": pm}
        if "SQ_LDS_BANK_CONFLICT" in pm and "SQ_LDS_IDX_ACTIVE" in pm:
            out["lds_bank_conflict_share_of_lds_cycles"] = pm["SQ_LDS_BANK_CONFLICT"]["mean"] / pm["SQ_LDS_IDX_ACTIVE"]["mean"]
        (dst / f"{tag}_c2_gcn2_swizzled_build_pmc.json").write_text(json.dumps(out, indent=2) + "\n")
