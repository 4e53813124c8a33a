#!/bin/bash
# Development: every gnn-builder_amd/libgnnb_v_*.so variant build through tests/zf_variants.py, twice (interleaved: boxes drift)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out/zfv
for rep in 1 2; do for lib in gnn-builder_amd/libgnnb_v_*.so; do t=$(basename $lib .so); t=${t#libgnnb_v_};
  GNNB_HIP_LIB=$R/$lib timeout 300 python3 tests/zf_variants.py $t 2>&1 | grep -E '^\{|Error|error|assert' ; done; done | tee gpurun_out/zfv/results.txt
