export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_ref_model_fixtures.py -m gpu -q -x -k "gin or deep or fused or stack or ref6 or whole_model or full_size" 2>&1 | grep -E "^E  .*assert|^E  |^FAILED|passed|failed" | head -20
for rep in 1 2; do
for w in c3 c3t ref6_gin ref6_gcn; do
  for f in 1 0; do
    GNNB_FUSED_SHAPE=$f python3 bench.py --workload $w --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$w shape$f', d['value'], d['ms_per_step'], d['roofline']['us_per_launch'], d['roofline']['frac'], d['config'].get('path'))"
  done
done
done
