export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -k "linear" 2>&1 | grep -E "^E  .*assert|^FAILED|passed|failed" | head
for rep in 1 2; do
for w in c4 c5 ref6_sage ref6_pna; do
    python3 bench.py --workload $w --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$w', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
done
done
