cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03o; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_roi_align.py tests/test_hip_pipeline.py -x -q -m gpu 2>&1 | tail -5 | tee $O/tests.txt
for v in "DTC_RA_TILE_DMA=0" "DTC_RA_TILE_DMA=1" "DTC_RA_TILE_DMA=1 DTC_RA_TILE_LDS_KB=40" "DTC_RA_TILE_DMA=1 DTC_RA_TILE_LDS_KB=78 DTC_RA_TILE_NQCAP=4" "DTC_RA_TILE_DMA=1 DTC_RA_TILE_LDS_KB=64" "DTC_RA_TILE_DMA=1 DTC_RA_TILE_MERGE=400"; do
  echo "== $v" | tee -a $O/summary.txt
  env $v timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'launch', {k: v for k, v in r.items() if 'launch' in k or k in ('frac','achieved')})" | tee -a $O/summary.txt
done
