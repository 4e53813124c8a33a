#!/usr/bin/env python3
"""Round-6 profile summaries for BASELINE configs 3-5: from one gpurun_out/<dir> produced by
    for w in c3 c4 c5:  rocprofv3 --kernel-trace --stats ... -- python3 bench.py --workload $w --roofline-only   (roofline_$w/)
                         rocprofv3 --pmc <group> ...          -- python3 bench.py --workload $w --roofline-only   (pmc_$w_{inst,busy,lds}/)
into profiles/r06_<w>_roofline_kernel_stats.csv (per-kernel launch statistics: every kernel of the timed step ALONE on the chip --
the roofline-only run's serial-forward leg -- beside the HBM-regime aggregate loops) and profiles/r06_<w>_kernels_pmc.json
(per kernel: instruction mix, matrix-pipe busy share, LDS bank conflicts, wait share; means per launch).
usage: summarize_r6.py gpurun_out/r6j profiles r06"""
import collections
import csv
import json
import sys
from pathlib import Path

src, dst, tag = Path(sys.argv[1]), Path(sys.argv[2]), sys.argv[3]
KERNELS = ("k_linear_dma", "k_pna_pagg", "k_pna_first", "k_sage_first_mean", "k_gcn2_fused", "k_gcn2_zf", "k_pool_mlp", "k_head_small",
           "k_pool_combine", "k_graph_prep", "k_aggregate_ring")
FLOPS = {  # algorithmic flops per launch of the dominant kernels at the benched shapes: 2 M K N
    ("c4", "k_linear_dma<0, 2>"): 2.0 * 147456 * 640 * 128, ("c5", "k_linear_dma<0, 1>"): 2.0 * 208896 * 512 * 256,
}


def short(name):
    n = name.split("(")[0]
    return n.replace("void ", "").replace("gnnb::", "").strip()


for w in ("c3", "c4", "c5"):
    st = list((src / f"roofline_{w}").rglob("roofline_kernel_stats.csv"))
    if not st:
        continue
    rows = list(csv.DictReader(open(st[0])))
    with open(dst / f"{tag}_{w}_roofline_kernel_stats.csv", "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload {w} --roofline-only  (N=1): the serial-forward leg (40 + 5 whole "
                f"forwards of one prepared batch on ONE stream: every kernel of the timed step alone on the chip), the stand-alone GEMM / k_pna_pagg "
                f"loops, the HBM-regime aggregate loops and their copy calibration (k_aggregate_ring<6>)\n")
        f.write("kernel,calls,total_us,avg_us,min_us,max_us,percent\n")
        for r in rows:
            f.write(f"\"{short(r['Name'])}\",{r['Calls']},{float(r['TotalDurationNs']) / 1e3:.1f},{float(r['AverageNs']) / 1e3:.2f},"
                    f"{float(r['MinNs']) / 1e3:.2f},{float(r['MaxNs']) / 1e3:.2f},{float(r['Percentage']):.2f}\n")
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in sorted(src.glob(f"pmc_{w}_*")):
        for p in sub.rglob("*counter_collection.csv"):
            for r in csv.DictReader(open(p)):
                k = short(r["Kernel_Name"])
                if any(k.startswith(x) for x in KERNELS):
                    per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    avg = {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in rows}
    out = {"command": f"rocprofv3 --pmc <group> -- python3 bench.py --workload {w} --roofline-only (one pass per counter group: "
                      "{SQ_INSTS_VALU MFMA SALU LDS} {SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES} "
                      "{SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU})",
           "workload": w, "kernels": {}}
    for k, cs in sorted(per.items()):
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        d = {"launches_counted": len(next(iter(cs.values()))), "avg_us_kernel_trace": avg.get(k), "counters_mean_per_launch": m}
        if "SQ_INSTS_VALU" in m and "SQ_INSTS_MFMA" in m:
            d["instruction_mix_per_launch"] = {"valu_non_mfma": m["SQ_INSTS_VALU"] - m["SQ_INSTS_MFMA"], "mfma": m["SQ_INSTS_MFMA"],
                                               "salu": m.get("SQ_INSTS_SALU"), "lds": m.get("SQ_INSTS_LDS")}
        if m.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_share_of_lds_cycles"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]
        if m.get("SQ_WAVE_CYCLES"):
            d["mfma_busy_cycles_per_simd"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0
            d["mfma_coexec_share"] = (m.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0.0) / m["SQ_VALU_MFMA_BUSY_CYCLES"]) if m.get("SQ_VALU_MFMA_BUSY_CYCLES") else None
            if avg.get(k):
                d["mfma_pipe_busy_share_of_kernel_at_2p4GHz"] = d["mfma_busy_cycles_per_simd"] / (avg[k] * 1e-6 * 2.4e9)
        if "SQ_WAIT_INST_ANY" in m and m.get("SQ_WAVE_CYCLES"):
            d["wait_inst_any_share_of_wave_cycles"] = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]
        fl = FLOPS.get((w, k))
        if fl and avg.get(k):
            d["algorithmic_flops_per_launch"] = fl
            d["frac_of_fp32_mfma_peak_157p3"] = fl / (avg[k] * 1e-6) / 157.3e12
        out["kernels"][k] = d
    (dst / f"{tag}_{w}_kernels_pmc.json").write_text(json.dumps(out, indent=1) + "\n")
    print(w, sorted(out["kernels"]))
