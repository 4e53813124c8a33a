#!/bin/bash
# Collect the round's rocprofv3 evidence on a GPU box (run through gpurun); outputs under gpurun_out/prof/.
#   tools/profile_r.sh [workload ...]      (default: c2 c3 c4 c5)
# Per workload: rocprofv3 --kernel-trace --stats of the bench command (its JSON line is kept beside the stats).
# For c2 also the --roofline-only loop (kernel trace, then one --pmc pass per counter group: counters are never
# collected together with trace domains other than kernel-trace).  The program comes directly after `--`.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
OUT=$R/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
WL=${@:-c2 c3 c4 c5}
for w in $WL; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_$w" -o bench -- python3 bench.py --workload $w --steps 100 > "$OUT/bench_$w.log" 2>&1
done
if echo "$WL" | grep -qw c2; then
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/roofline" -o roofline -- python3 bench.py --roofline-only > "$OUT/roofline.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- python3 bench.py --roofline-only > "$OUT/pmc_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o write -- python3 bench.py --roofline-only > "$OUT/pmc_write.log" 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_l2" -o l2 -- python3 bench.py --roofline-only > "$OUT/pmc_l2.log" 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_inst" -o inst -- python3 bench.py --roofline-only > "$OUT/pmc_inst.log" 2>&1
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d "$OUT/pmc_busy" -o busy -- python3 bench.py --roofline-only > "$OUT/pmc_busy.log" 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc_wait" -o wait -- python3 bench.py --roofline-only > "$OUT/pmc_wait.log" 2>&1
fi
# keep the merge small: the per-dispatch traces are not needed once the stats exist
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*.csv" | xargs ls -la
tail -c 600 "$OUT"/bench_*.log
