#!/bin/bash
# NEEDS commit 1e29752 plus the depth-2 variant described in tools/r05/README.md 4c (not kept); the record of the commands
# round 5, call 7: tile2 with the loads two passes ahead (8 pieces in flight, 52 KB, 3 workgroups / CU) against the shipped kernel
DTC_RA_TILE2=1 DTC_RA_TILE2_DEPTH=2 python -m pytest tests/test_hip_roi_align.py -x -q -m gpu -k "window_shapes or edge_cases or full_channel or (real_shape and fp32 and nchw) or golden" 2>&1 | tail -2
for rep in 1 2; do
for cfg in "DTC_RA_TILE2=0" "DTC_RA_TILE2=1 DTC_RA_TILE2_DEPTH=2" "DTC_RA_TILE2=1 DTC_RA_TILE2_DEPTH=2 DTC_RA_TILE2_LDS_KB=40" "DTC_RA_TILE2=1 DTC_RA_TILE2_DEPTH=1 DTC_RA_TILE2_LDS_KB=52"; do
  echo "== $cfg"
  env $cfg python tools/bench_boxhead.py --iters 40
  env $cfg python tools/bench_boxhead.py --iters 40 --harder
  env $cfg python tools/bench_boxhead.py --iters 40 --mask
done; done
bash tools/r05/counters.sh r05_t2d2 DTC_RA_TILE2=1 DTC_RA_TILE2_DEPTH=2 > gpurun_out/r05_t2d2.txt 2>&1; grep -E "^(l1_|l2_|launch_ms|TCC_EA0_RDREQ_sum|SQ_INSTS_VALU |    )" gpurun_out/r05_t2d2.txt
