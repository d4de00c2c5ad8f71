#!/bin/bash
# NEEDS the kernel of commit 1e29752 (git show 1e29752:detectorch_amd/csrc/roi_align_tile.hip): roi_align_fwd_tile2 is not in the product
# round 5, call 1: the restructured float32 cluster kernel (tile2): parity tests, then A/B against the round-4 kernel in one call
python -m pytest tests/test_hip_roi_align.py -x -q -m gpu -k "tile2 or real_shape or edge_cases or full_channel or golden or fpn_multilevel or special" 2>&1 | tail -8
for rep in 1 2; do
for v in 1 0; do
  echo "== DTC_RA_TILE2=$v"
  DTC_RA_TILE2=$v python tools/bench_boxhead.py --iters 40
  DTC_RA_TILE2=$v python tools/bench_boxhead.py --iters 40 --harder
  DTC_RA_TILE2=$v python tools/bench_boxhead.py --iters 40 --mask
done; done
