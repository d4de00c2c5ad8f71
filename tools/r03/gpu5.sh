cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03h; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_roi_align.py tests/test_hip_pipeline.py -x -q -m gpu 2>&1 | tail -2 | tee -a $O/summary.txt
for e in "DTC_RA_MAP_PREP=1" "DTC_RA_MAP_PREP=0"; do env $e timeout 300 python bench.py --workload cfg2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$e cfg2:', d['value'], 'img/s', d['ms_per_step'], 'ms/step; launch', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])" | tee -a $O/summary.txt; done
timeout 300 python bench.py --workload cfg2 --c4-pooled 14 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg2 14x14:', d['value'], 'img/s', d['ms_per_step'], 'ms/step; launch', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])" | tee -a $O/summary.txt
