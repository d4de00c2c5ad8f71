cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03k; mkdir -p $O
for kv in "X=0" "DTC_RA_TILE_PHASE=100" "DTC_RA_TILE_PHASE=200" "DTC_RA_TILE_PHASE=400" "DTC_RA_TILE_PHASE=800" "DTC_RA_TILE_PHASE=1600" "DTC_RA_TILE_PHASE=400,DTC_FPN_BAND_LOG2=5"; do
  echo -n "$kv : " | tee -a $O/summary.txt; env ${kv//,/ } timeout 200 python tools/bench_boxhead.py --batch 8 --iters 30 2>/dev/null | tail -1 | tee -a $O/summary.txt
done
DTC_RA_TILE_PHASE=400 timeout 600 python -m pytest tests/test_hip_pipeline.py tests/test_hip_roi_align.py -x -q -m gpu 2>&1 | tail -2 | tee -a $O/summary.txt
