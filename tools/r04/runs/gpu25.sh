#!/bin/bash
# bf16 NCHW maps through the cluster kernel: 3 waves / SIMD without spills (ab_old) against 4 with 12-40 B of scratch per lane
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OLD=$PWD/detectorch_amd/lib/ab_old/libdetectorch_hip.so
for rep in 1 2; do
for A in "--bf16" "--bf16 --top-n 2000" "--bf16 --mask" "--fp16"; do
  echo -n "3 waves $A | "; DETECTORCH_HIP_LIB=$OLD timeout 300 python tools/bench_boxhead.py $A 2>&1 | tail -1
  echo -n "4 waves $A | "; timeout 300 python tools/bench_boxhead.py $A 2>&1 | tail -1
done; done
timeout 1500 python -m pytest tests/test_hip_roi_align.py -m gpu -x -q 2>&1 | tail -2
