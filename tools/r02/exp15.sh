#!/bin/bash
# workgroup ordering inside an XCD slice: channel-major sub-slices of S groups (DTC_RA_TILE_ORDER=S)
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
B="python tools/bench_boxhead.py"
for cfg in "X=1" "DTC_RA_TILE_ORDER=12" "DTC_RA_TILE_ORDER=24" "DTC_RA_TILE_ORDER=48" "DTC_RA_TILE_ORDER=96" "DTC_RA_TILE_ORDER=400" "DTC_RA_TILE_ORDER=24 DTC_RA_TILE_CHBLOCK=32" "DTC_RA_TILE_ORDER=48 DTC_RA_TILE_CHBLOCK=32" "DTC_RA_TILE_ORDER=96 DTC_RA_TILE_CHBLOCK=32" "DTC_RA_TILE_ORDER=96 DTC_RA_TILE_CHBLOCK=16" "DTC_RA_TILE_ORDER=48 DTC_RA_TILE_CHBLOCK=128" "X=2"; do
  echo -n "$cfg : "; env $cfg timeout 200 $B 2>&1 | tail -1
done
echo mask
for cfg in "X=1" "DTC_RA_TILE_ORDER=24" "DTC_RA_TILE_ORDER=96"; do
  echo -n "$cfg : "; env $cfg timeout 200 $B --mask 2>&1 | tail -1
done
