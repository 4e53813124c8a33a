export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "pna_degree_classes" 2>&1 | grep -E "^E  |^FAILED|passed|failed|Error" | head -30
cd /tmp
for w in c4; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks/$w -o b -- python3 $R/bench.py --workload $w --streams 1 --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-roofline > $R/gpurun_out/ks/$w.log 2>&1
f=$(find $R/gpurun_out/ks/$w -name "*kernel_stats.csv" | head -1)
python3 - "$f" $w <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]: print(sys.argv[2], r['Name'][:90], r['Calls'], round(float(r['AverageNs'])/1000,1), r['Percentage'])
PY
done
find $R/gpurun_out/ks -name "*kernel_trace.csv" -delete
cd $R
for rep in 1 2; do for w in c4 ref6_pna; do
    python3 bench.py --workload $w --steps 50 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$w', d['value'], d['ms_per_step'])"
done; done
