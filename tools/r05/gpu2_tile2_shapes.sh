#!/bin/bash
# NEEDS the kernel of commit 1e29752 (git show 1e29752:detectorch_amd/csrc/roi_align_tile.hip): roi_align_fwd_tile2 is not in the product
# round 5, call 2: tile2 shapes (256 / 512 threads, 4x4 / 2x8 blocks): parity, then timing A/B against the round-4 kernel
for cfg in "DTC_RA_TILE2_NT=256 DTC_RA_TILE2_BLOCK=2" "DTC_RA_TILE2_NT=256 DTC_RA_TILE2_BLOCK=3" "DTC_RA_TILE2_NT=512 DTC_RA_TILE2_BLOCK=2" "DTC_RA_TILE2_NT=512 DTC_RA_TILE2_BLOCK=3"; do
  echo "== tests $cfg"; env $cfg python -m pytest tests/test_hip_roi_align.py -x -q -m gpu -k "tile2 or edge_cases or full_channel or (real_shape and fp32 and nchw)" 2>&1 | tail -3
done
for rep in 1 2; do
for cfg in "DTC_RA_TILE2=0" "DTC_RA_TILE2_NT=256 DTC_RA_TILE2_BLOCK=2" "DTC_RA_TILE2_NT=256 DTC_RA_TILE2_BLOCK=3" "DTC_RA_TILE2_NT=512 DTC_RA_TILE2_BLOCK=2" "DTC_RA_TILE2_NT=512 DTC_RA_TILE2_BLOCK=3"; do
  echo "== $cfg"
  env $cfg python tools/bench_boxhead.py --iters 40
  env $cfg python tools/bench_boxhead.py --iters 40 --harder
  env $cfg python tools/bench_boxhead.py --iters 40 --mask
done; done
