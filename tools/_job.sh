export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2; do
for w in c4 c5 ref6_sage ref6_pna; do
  for ts in 2 1; do
    GNNB_GEMM_TAIL_SPLIT=$ts python3 bench.py --workload $w --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$w ts$ts', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
  done
done
done
timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -q -k "full_size or pooling or pna or sage" 2>&1 | tail -3
