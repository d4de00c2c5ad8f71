#!/bin/bash
# NEEDS the kernel of commit 1e29752 (git show 1e29752:detectorch_amd/csrc/roi_align_tile.hip): roi_align_fwd_tile2 is not in the product
# round 5, call 3: channel-block size x kernel (shorter-lived workgroups stay in phase with their neighbours: more L2 reuse)
for rep in 1 2; do
for cfg in "DTC_RA_TILE2=0" "DTC_RA_TILE2=0 DTC_RA_TILE_CHBLOCK=32" "DTC_RA_TILE2_NT=256" "DTC_RA_TILE2_NT=256 DTC_RA_TILE_CHBLOCK=32" "DTC_RA_TILE2_NT=256 DTC_RA_TILE_CHBLOCK=16" "DTC_RA_TILE2_NT=512 DTC_RA_TILE_CHBLOCK=32" "DTC_RA_TILE2_NT=512 DTC_RA_TILE_CHBLOCK=16"; do
  echo "== $cfg"
  env $cfg python tools/bench_boxhead.py --iters 40
  env $cfg python tools/bench_boxhead.py --iters 40 --harder
done; done
