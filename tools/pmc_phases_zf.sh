#!/bin/bash
# Development only (library built with -DGNNB_ZF_ABLATE): instruction counts of k_gcn2_zf with phases switched off, one
# rocprofv3 --pmc pass per variant (program directly after `--`): which phase issues how many VALU / SALU / LDS instructions.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
OUT=$R/gpurun_out/pmc_phases
rm -rf "$OUT"; mkdir -p "$OUT"
for dbg in ${ZF_VARIANTS:-0 1 4 8 16 12 19 31 64}; do   # (never 2 alone: P1 on garbage records can loop for seconds)
  GNNB_ZF_DBG=$dbg GNNB_ZF_SHAPE=${ZF_SHAPE:-0} rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$OUT/d$dbg" -o p -- python3 tools/few_zf.py 6 > "$OUT/d$dbg.log" 2>&1
done
python3 - <<'PY'
import csv, glob, collections
names = {0: "everything", 1: "no P1", 2: "no P0'", 4: "no M1", 8: "no M0", 16: "no Z write", 12: "no MFMA phases", 19: "MFMA phases only", 31: "skeleton", 64: "return after first DMA"}
base = None
import os
for dbg in [int(v) for v in os.environ.get('ZF_VARIANTS', '0 1 4 8 16 12 19 31 64').split()]:
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_phases/d{dbg}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gcn2_zf" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    if not m:
        print(dbg, "no data"); continue
    valu = m["SQ_INSTS_VALU"] - m["SQ_INSTS_MFMA"]
    if base is None:
        base = (valu, m["SQ_INSTS_MFMA"], m["SQ_INSTS_SALU"], m["SQ_INSTS_LDS"])
    print(f"dbg {dbg:2d} {names[dbg]:24s} VALU {valu/1e6:6.2f} M  MFMA {m['SQ_INSTS_MFMA']/1e6:5.2f} M  SALU {m['SQ_INSTS_SALU']/1e6:5.2f} M  LDS {m['SQ_INSTS_LDS']/1e6:5.2f} M   "
          f"(delta VALU {(base[0]-valu)/1e6:5.2f}, SALU {(base[2]-m['SQ_INSTS_SALU'])/1e6:5.2f}, LDS {(base[3]-m['SQ_INSTS_LDS'])/1e6:5.2f})")
PY
