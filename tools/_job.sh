export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -k "pooling_in_the_last or full_size_configs_3_4_5 or pna_lin_folded" 2>&1 | tail -1
for rep in 1 2; do for w in c5 ref6_sage; do
    python3 bench.py --workload $w --steps 50 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$w', d['value'], d['ms_per_step'])"
done; done
