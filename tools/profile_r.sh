#!/bin/bash
# Collect the round's rocprofv3 evidence on a GPU box (run through gpurun); outputs under gpurun_out/prof/.
#   tools/profile_r.sh [workloads ...]      (default: c2 c3 c3t c4 c5)
# Per workload w:
#   plain_$w.log  `python3 bench.py --workload w --steps 100` WITHOUT the profiler: the JSON line that is kept (under the
#                 profiler the back-to-back HIP-event loops of the roofline legs run up to 2x slower -- its per-dispatch
#                 instrumentation -- so the line printed there is not a measurement of them)
#   bench_$w/     rocprofv3 --kernel-trace --stats of the same command: the per-kernel table
#   roofline_$w/  the same of `bench.py --workload w --roofline-only` (HBM-regime gather-aggregate loop, its copy
#                 calibration, the conv-stack loop where the workload has one)
#   pmc_$w_*/     one --pmc pass per counter group over the roofline-only command (counters are never collected together
#                 with trace domains other than kernel-trace; the program comes directly after `--`)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
OUT=$R/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
WL=${@:-c2 c3 c4 c5}
for w in $WL; do
  extra="--no-other-configs"   # (every tracked line keeps its cpu_baseline; the brief c3 / c4 / c5 legs of the default line would mix their kernels into the c2 table)
  python3 bench.py --workload $w --steps 100 $extra > "$OUT/plain_$w.log" 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_$w" -o bench -- python3 bench.py --workload $w --steps 100 $extra > "$OUT/bench_$w.log" 2>&1
  case $w in ref6_*) continue;; esac   # (the published-model workloads: the line and the kernel table only)
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/roofline_$w" -o roofline -- python3 bench.py --workload $w --roofline-only > "$OUT/roofline_$w.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_${w}_fetch" -o fetch -- python3 bench.py --workload $w --roofline-only > "$OUT/pmc_${w}_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_${w}_write" -o write -- python3 bench.py --workload $w --roofline-only > "$OUT/pmc_${w}_write.log" 2>&1
  if [ "$w" = "c3" ] || [ "$w" = "c3t" ]; then
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_${w}_inst" -o inst -- python3 bench.py --workload $w --roofline-only > "$OUT/pmc_${w}_inst.log" 2>&1
    rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d "$OUT/pmc_${w}_busy" -o busy -- python3 bench.py --workload $w --roofline-only > "$OUT/pmc_${w}_busy.log" 2>&1
  fi
  if [ "$w" = "c2" ]; then
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_${w}_l2" -o l2 -- python3 bench.py --workload $w --roofline-only > "$OUT/pmc_${w}_l2.log" 2>&1
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_${w}_inst" -o inst -- python3 bench.py --workload $w --roofline-only > "$OUT/pmc_${w}_inst.log" 2>&1
    rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d "$OUT/pmc_${w}_busy" -o busy -- python3 bench.py --workload $w --roofline-only > "$OUT/pmc_${w}_busy.log" 2>&1
    rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc_${w}_wait" -o wait -- python3 bench.py --workload $w --roofline-only > "$OUT/pmc_${w}_wait.log" 2>&1
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/pmc_${w}_lds" -o lds -- python3 bench.py --workload $w --roofline-only > "$OUT/pmc_${w}_lds.log" 2>&1
  fi
done
# k_gcn2_zf built with its H / Z rows XOR-swizzled (-DZF_SWZ=1, gnn-builder_amd/libgnnb_v_swz.so when present): the LDS bank
# conflicts of the shipped kernel's fragment reads, gone -- and what that costs
if [ -f "$R/gnn-builder_amd/libgnnb_v_swz.so" ]; then
  GNNB_HIP_LIB=$R/gnn-builder_amd/libgnnb_v_swz.so python3 bench.py --workload c2 --roofline-only > "$OUT/swz_roofline.log" 2>&1
  GNNB_HIP_LIB=$R/gnn-builder_amd/libgnnb_v_swz.so rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/pmc_swz_lds" -o lds -- python3 bench.py --workload c2 --roofline-only > "$OUT/pmc_swz_lds.log" 2>&1
  GNNB_HIP_LIB=$R/gnn-builder_amd/libgnnb_v_swz.so rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_swz_inst" -o inst -- python3 bench.py --workload c2 --roofline-only > "$OUT/pmc_swz_inst.log" 2>&1
fi
# keep the merge small: the per-dispatch traces are not needed once the stats exist
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*.csv" | xargs ls -la | head -60
tail -c 400 "$OUT"/plain_*.log
