#!/bin/bash
# Counters of the conv-stack kernels (k_gcn2_zf / k_gcn2_fused) on the BASELINE config 2 batch: one rocprofv3 --pmc pass per
# counter group (never combined with trace domains), program directly after `--`.  Output: gpurun_out/pmc_stack/summary.json
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
OUT=$R/gpurun_out/pmc_stack
rm -rf "$OUT"; mkdir -p "$OUT"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/p$i" -o p -- python3 tools/time_gcn2.py > "$OUT/p$i.log" 2>&1
done
python3 - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_stack/*/*counter_collection.csv")) + sorted(glob.glob("gpurun_out/pmc_stack/*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "gcn2" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: {"mean": sum(v) / len(v), "launches": len(v)} for c, v in d.items()} for k, d in agg.items()}
json.dump(out, open("gpurun_out/pmc_stack/summary.json", "w"), indent=1)
for k, d in out.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} {v['mean']:16.1f}  (n={v['launches']})")
PY
