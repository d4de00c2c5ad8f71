cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r03check}; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest -m gpu rc $?" | tee -a $O/summary.txt; grep -E "passed|failed" $O/pytest.log | tail -2 | tee -a $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a $O/summary.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cat $O/bench_default.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('default: value', d['value'], 'ms/step', d['ms_per_step'], 'launch', r['avg_launch_ms'], r['launch_ms_min_median_max'], 'frac', r['frac'], 'traffic', r['traffic'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['kind'])" | tee -a $O/summary.txt
