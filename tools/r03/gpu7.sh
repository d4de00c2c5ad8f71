cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03j; mkdir -p $O
for kv in "X=0" "DTC_FPN_GROUP=5" "DTC_FPN_GROUP=5,DTC_RA_TILE_REVERSE=0" "DTC_FPN_GROUP=10" "DTC_FPN_GROUP=5,DTC_FPN_BAND_LOG2=3"; do
  echo -n "$kv : " | tee -a $O/summary.txt; env ${kv//,/ } timeout 200 python tools/bench_boxhead.py --batch 8 --iters 30 2>/dev/null | tail -1 | tee -a $O/summary.txt
done
DTC_FPN_GROUP=5 timeout 600 python -m pytest tests/test_hip_pipeline.py tests/test_hip_fpn_det_mask.py -x -q -m gpu 2>&1 | tail -2 | tee -a $O/summary.txt
