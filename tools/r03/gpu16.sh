cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03u; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_epilogue.py -x -q -m gpu 2>&1 | tail -25 | tee $O/tests.txt
for args in "--dtype bf16 --channels-last" "--dtype bf16 --channels-last --optimize" "--dtype bf16 --optimize" "--dtype fp32 --optimize"; do
  timeout 600 python tools/bench_detector.py --batched --batch 8 $args 2>&1 | tail -1 | cut -c1-330 | tee -a $O/summary.txt
done
timeout 600 python tools/r03/prof_batched.py opt > $O/prof_opt.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'launch', r['avg_launch_ms'], r['launch_ms_min_median_max'], r['launch_ms_samples'])" | tee -a $O/summary.txt
