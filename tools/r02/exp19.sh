#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
for cfg in "DTC_RA_TILE_REVERSE=0" "DTC_RA_TILE_REVERSE=1" "DTC_RA_TILE_REVERSE=0" "DTC_RA_TILE_REVERSE=1"; do
  echo -n "$cfg : "; env $cfg timeout 200 python tools/bench_boxhead.py 2>&1 | tail -1
done
for cfg in "DTC_RA_TILE_REVERSE=0" "DTC_RA_TILE_REVERSE=1"; do
  echo -n "$cfg : "; env $cfg timeout 200 python tools/bench_boxhead.py --mask 2>&1 | tail -1
  echo -n "$cfg : "; env $cfg timeout 200 python tools/bench_roialign.py --sort 2>&1 | tail -1
done
timeout 200 python tools/r02/trace_tile.py 2>&1 | grep -v amdgpu.ids | head -24
