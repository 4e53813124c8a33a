export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2; do for v in 32 16; do for w in c4 ref6_pna; do
    GNNB_SK_RC=$v python3 bench.py --workload $w --steps 50 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$w sk_rc=$v', d['value'], d['ms_per_step'])"
done; done; done
GNNB_SK_RC=16 timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -k "pna_degree" 2>&1 | tail -1
