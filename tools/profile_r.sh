#!/bin/bash
# Collect the round's rocprofv3 evidence on a GPU box (run through gpurun); outputs under gpurun_out/prof/.
# Counters are collected in their own passes (never together with trace domains other than kernel-trace).
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
OUT=$R/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -o bench -- python3 bench.py > "$OUT/bench.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/roofline" -o roofline -- python3 bench.py --roofline-only > "$OUT/roofline.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- python3 bench.py --roofline-only > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o write -- python3 bench.py --roofline-only > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_l2" -o l2 -- python3 bench.py --roofline-only > "$OUT/pmc_l2.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_inst" -o inst -- python3 bench.py --roofline-only > "$OUT/pmc_inst.log" 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d "$OUT/pmc_busy" -o busy -- python3 bench.py --roofline-only > "$OUT/pmc_busy.log" 2>&1
find "$OUT" -name "*.csv" | xargs ls -la
tail -2 "$OUT"/*.log
