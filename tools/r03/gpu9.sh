cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03m; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_roi_align.py -x -q -m gpu -k "nhwc or channels_last or bfloat16 or ragged" 2>&1 | tail -3 | tee -a $O/summary.txt
for kv in "X=0" "DTC_RA_NHWC_LDS=0" "DTC_RA_NHWC_LDS_KB=52" "DTC_RA_NHWC_LDS_KB=104" "DTC_RA_NHWC_LDS_KB=156"; do
  echo -n "$kv fp32: " | tee -a $O/summary.txt; env $kv timeout 200 python tools/bench_roialign.py --nhwc --sort --max-side 90 2>/dev/null | tail -1 | tee -a $O/summary.txt
done
for kv in "X=0" "DTC_RA_NHWC_LDS=0"; do
  echo -n "$kv fp16: " | tee -a $O/summary.txt; env $kv timeout 200 python tools/bench_roialign.py --nhwc --half --sort --max-side 90 2>/dev/null | tail -1 | tee -a $O/summary.txt
  echo -n "$kv bench --channels-last: " | tee -a $O/summary.txt; env $kv timeout 300 python bench.py --channels-last --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'img/s', d['ms_per_step'], 'ms/step; box launch', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])" | tee -a $O/summary.txt
done
echo -n "NCHW reference point: " | tee -a $O/summary.txt; timeout 200 python tools/bench_roialign.py --sort --max-side 90 2>/dev/null | tail -1 | tee -a $O/summary.txt
