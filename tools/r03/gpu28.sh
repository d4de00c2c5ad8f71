cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03ah; mkdir -p $O
for v in "DTC_RA_NHWC_LDS=1" "DTC_RA_NHWC_LDS=0" "DTC_RA_NHWC_LDS=0 DTC_RA_NHWC_F32_CB32=1"; do
echo "== $v" | tee -a $O/summary.txt
env $v timeout 300 python bench.py --channels-last --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('bench --channels-last (fp32): value', d['value'], 'ms/step', d['ms_per_step'], 'launch', r['avg_launch_ms'], r['launch_ms_min_median_max'])" | tee -a $O/summary.txt
done
