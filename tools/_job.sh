export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 1800 python tests/fuzz_layerwise.py 120 21 2>&1 | grep -E "pna|cases|FAIL" | tail -14
