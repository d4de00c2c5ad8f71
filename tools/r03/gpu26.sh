cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03af; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_roi_align.py -x -q -m gpu 2>&1 | tail -4 | tee $O/tests.txt
for a in "--nhwc --half --sort --max-side 90" "--nhwc --half --sort"; do echo -n "$a: " | tee -a $O/summary.txt; timeout 200 python tools/bench_roialign.py $a 2>/dev/null | tail -1 | tee -a $O/summary.txt; done
timeout 300 python bench.py --channels-last --fp16 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('bench --channels-last --fp16: value', d['value'], 'ms/step', d['ms_per_step'], 'launch', r['avg_launch_ms'], r['launch_ms_min_median_max'])" | tee -a $O/summary.txt
