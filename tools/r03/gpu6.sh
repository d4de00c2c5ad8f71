cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03i; mkdir -p $O
for b in 8 16; do for kv in "X=0" "DTC_RA_TILE_IMGGROUPS=200" "DTC_RA_TILE_IMGGROUPS=200,DTC_RA_TILE_REVERSE=0" "DTC_RA_TILE_REVERSE=0"; do
  echo -n "batch $b $kv : " | tee -a $O/summary.txt; env ${kv//,/ } timeout 200 python tools/bench_boxhead.py --batch $b --iters 30 2>/dev/null | tail -1 | tee -a $O/summary.txt
done; done
