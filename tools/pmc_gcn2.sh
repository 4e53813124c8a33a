#!/bin/bash
# Instruction-mix counters of the fused GCN kernel (own rocprofv3 pass, --pmc only).
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
OUT=$R/gpurun_out/pmc_g2
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$OUT/p1" -o p1 -- python3 bench.py --streams 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > "$OUT/p1.log" 2>&1
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_g2/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        if "gcn2" in k:
            print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
