export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 1500 python tests/fuzz_layerwise.py 80 1 2>&1 | tail -12
timeout 900 python tests/fuzz_fused.py 60 7 2>&1 | tail -3
timeout 600 python tests/fuzz_fused.py 40 8 zf 2>&1 | tail -2
timeout 900 python tools/fuzz_gemm.py 120 11 2>&1 | tail -2
