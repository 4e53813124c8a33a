export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_ref_model_fixtures.py -m gpu -q -k "pna or aggregate" 2>&1 | grep -E "^E  |^FAILED|passed|failed|Error" | head -20
timeout 900 python tests/fuzz_layerwise.py 60 51 2>&1 | tail -1
for rep in 1 2; do for w in c4 ref6_pna; do
    python3 bench.py --workload $w --steps 50 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$w', d['value'], d['ms_per_step'])"
done; done
