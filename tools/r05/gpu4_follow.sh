#!/bin/bash
# NEEDS the reverted experiment code (a global progress table + pick() in tile_passes: tools/r05/README.md 4b); kept as the record of the commands
# round 5, call 4: follow-the-predecessor pass order (DTC_RA_TILE_FOLLOW=1) against the shipped order: parity, times, L2 counters
DTC_RA_TILE_FOLLOW=1 python -m pytest tests/test_hip_roi_align.py -x -q -m gpu -k "window_shapes or edge_cases or full_channel or (real_shape and fp32 and nchw) or golden" 2>&1 | tail -2
for rep in 1 2; do for f in 0 1; do
  echo "== FOLLOW=$f"; DTC_RA_TILE_FOLLOW=$f python tools/bench_boxhead.py --iters 40; DTC_RA_TILE_FOLLOW=$f python tools/bench_boxhead.py --iters 40 --harder
done; done
for f in 0 1; do bash tools/r05/counters.sh r05_follow$f DTC_RA_TILE_FOLLOW=$f > gpurun_out/r05_follow$f.txt 2>&1; grep -E "^(l1_|l2_|launch_ms|TCC_EA0_RDREQ_sum)" gpurun_out/r05_follow$f.txt; done
