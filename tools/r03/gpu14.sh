cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03r; mkdir -p $O
timeout 300 python tools/r03/merged_check.py 2>&1 | tail -12 | tee $O/check.txt
for v in "DTC_RA_EXACT=1" "DTC_RA_EXACT=0" "DTC_RA_EXACT=0 DTC_RA_TILE_DMA=1 DTC_RA_TILE_LDS_KB=40"; do
  echo "== $v" | tee -a $O/summary.txt
  env $v timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'launch', r['avg_launch_ms'], r['launch_ms_min_median_max'])" | tee -a $O/summary.txt
done
