#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
B="python tools/bench_boxhead.py"
for cfg in "DTC_RA_TILE_ABLATE=15" "DTC_RA_TILE_ABLATE=143" "DTC_RA_TILE_ABLATE=271" "DTC_RA_TILE_ABLATE=64" "DTC_RA_TILE_ABLATE=15 DTC_RA_TILE_LDS_KB=160" "DTC_RA_TILE_ABLATE=15 DTC_RA_TILE_LDS_KB=78" "DTC_RA_TILE_ABLATE=0 DTC_RA_TILE_LDS_KB=78"; do
  echo -n "$cfg : "; env DTC_FPN_BAND_LOG2=4 DTC_RA_TILE_CHBLOCK=64 $cfg timeout 200 $B 2>&1 | tail -1
done
