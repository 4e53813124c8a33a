#!/bin/bash
# Counter passes for one gnnb_linear shape (own rocprofv3 runs, --pmc only).  tools/pmc_gemm.sh M N K [kernel-name-part]
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
M=${1:-147456}; N=${2:-128}; K=${3:-1664}; PAT=${4:-k_linear}
OUT=$R/gpurun_out/pmc_gemm
rm -rf "$OUT"; mkdir -p "$OUT"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
    i=$((i + 1))
    # (FETCH_SIZE / WRITE_SIZE together in one pass aborted rocprofv3 on this pool and hung until the limit: left out;
    #  every pass under its own timeout)
    timeout 180 rocprofv3 --pmc $grp --output-format csv -d "$OUT/p$i" -o p$i -- python3 tools/run_gemm_once.py $M $N $K > "$OUT/p$i.log" 2>&1
done
python3 - "$PAT" "$M" "$N" "$K" <<'PY'
import csv, glob, collections, json, sys
pat = sys.argv[1]
out = {}
for f in sorted(glob.glob("gpurun_out/pmc_gemm/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        if pat in k:
            for c, v in d.items():
                out.setdefault(k, {})[c] = {"mean": sum(v) / len(v), "launches": len(v)}
print(json.dumps({"shape": sys.argv[2:5], "counters_per_launch": out}, indent=1))
PY
