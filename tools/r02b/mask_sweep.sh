#!/bin/bash
# mask-head (14x14) RoIAlign launch of the bench workload under the cluster kernel's development knobs
for env in "" "DTC_RA_TILE_CHBLOCK=128" "DTC_RA_TILE_CHBLOCK=256" "DTC_RA_TILE_CHBLOCK=32" "DTC_RA_TILE_NQCAP=2" "DTC_RA_TILE_NQCAP=8" "DTC_RA_TILE_REVERSE=0" "DTC_RA_TILE_LDS_KB=38"; do
  echo -n "[$env] "; env $env timeout 100 python tools/bench_boxhead.py --mask --iters 50 2>&1 | tail -1
done
