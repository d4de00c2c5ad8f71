#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
B="python tools/bench_boxhead.py --mask"
for cfg in "X=1" "DTC_RA_TILE_CHBLOCK=32" "DTC_RA_TILE_CHBLOCK=16" "DTC_RA_TILE_CHBLOCK=128" "DTC_RA_TILE_NQCAP=2" "DTC_RA_TILE_NQCAP=1" "DTC_RA_TILE_LDS_KB=78" "DTC_RA_TILE_LDS_KB=39" "DTC_RA_TILE_LDS_KB=39 DTC_RA_TILE_CHBLOCK=32" "DTC_ROIALIGN_TILE=0"; do
  echo -n "$cfg : "; env $cfg timeout 200 $B 2>&1 | tail -1
done
