export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -3
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python3 bench.py 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('c2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])"
