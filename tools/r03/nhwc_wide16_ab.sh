cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03ag; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_roi_align.py tests/test_hip_detector.py -x -q -m gpu 2>&1 | tail -4 | tee $O/tests.txt
for v in "DTC_RA_NHWC_WIDE16=0" "DTC_RA_NHWC_WIDE16=1"; do
echo "== $v" | tee -a $O/summary.txt
env $v timeout 300 python bench.py --channels-last --fp16 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('bench --channels-last --fp16: value', d['value'], 'ms/step', d['ms_per_step'], 'launch', r['avg_launch_ms'], r['launch_ms_min_median_max'])" | tee -a $O/summary.txt
env $v timeout 200 python tools/bench_roialign.py --nhwc --half --sort --pooled 14 --rois 100 2>/dev/null | tail -1 | tee -a $O/summary.txt
done
timeout 900 python tools/bench_detector.py --batched --batch 8 --dtype bf16 --channels-last --optimize --miopen-benchmark 2>&1 | tail -1 | cut -c1-360 | tee -a $O/summary.txt
