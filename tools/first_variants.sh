#!/bin/bash
# Development: k_conv_first variant builds (gnn-builder_amd/libgnnb_v_*.so) timed inside the C5 forward (rocprofv3 kernel stats)
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out/kfv; cd /tmp
for lib in $R/gnn-builder_amd/libgnnb_v_*.so; do t=$(basename $lib .so); t=${t#libgnnb_v_}
  GNNB_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kfv/$t -o b -- python3 $R/bench.py --workload ${1:-c5} --streams 1 --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-roofline > $R/gpurun_out/kfv/$t.log 2>&1
  f=$(find $R/gpurun_out/kfv/$t -name "*kernel_stats.csv" | head -1)
  echo "$t $(grep -E 'k_conv_first|k_linear_reg<2' $f | awk -F, '{printf "%s avg %.1f min %.1f ", substr($1,1,40), $4/1000, $6/1000}')"
done
find $R/gpurun_out/kfv -name "*kernel_trace.csv" -delete
