#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02_exp4; mkdir -p $O
export TMPDIR=/tmp
B="python tools/bench_boxhead.py"
echo "== timings" | tee $O/times.log
for cfg in "DTC_RA_TILE_ABLATE=16" "DTC_RA_TILE_ABLATE=32" "DTC_RA_TILE_ABLATE=64" "DTC_RA_TILE_ABLATE=15" "DTC_RA_TILE_ABLATE=0" \
   "DTC_RA_TILE_ABLATE=16 DTC_RA_TILE_CHBLOCK=256" "DTC_RA_TILE_ABLATE=32 DTC_RA_TILE_CHBLOCK=256" "DTC_RA_TILE_ABLATE=64 DTC_RA_TILE_CHBLOCK=256" "DTC_RA_TILE_ABLATE=15 DTC_RA_TILE_CHBLOCK=256"; do
  echo -n "$cfg : " | tee -a $O/times.log; env DTC_FPN_BAND_LOG2=4 DTC_RA_TILE_CHBLOCK=64 $cfg timeout 200 $B 2>&1 | tail -1 | tee -a $O/times.log
done
