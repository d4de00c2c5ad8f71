#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
B="python tools/bench_boxhead.py"
for cfg in "DTC_FPN_GROUP_LOG2=0" "DTC_FPN_GROUP_LOG2=1" "DTC_FPN_GROUP_LOG2=2" "DTC_FPN_GROUP_LOG2=1 DTC_FPN_BAND_LOG2=3" "DTC_FPN_GROUP_LOG2=2 DTC_FPN_BAND_LOG2=3" "DTC_FPN_GROUP_LOG2=1 DTC_FPN_BAND_LOG2=5" \
           "DTC_FPN_GROUP_LOG2=1 DTC_RA_TILE_CHBLOCK=128" "DTC_FPN_GROUP_LOG2=1 DTC_RA_TILE_NT=512" "DTC_FPN_GROUP_LOG2=1 DTC_RA_TILE_MERGE=400"; do
  echo -n "$cfg : "; env $cfg timeout 200 $B 2>&1 | tail -1
  echo -n "   mask: "; env $cfg timeout 200 $B --mask 2>&1 | tail -1
done
timeout 300 python -m pytest tests/test_hip_fpn_det_mask.py tests/test_hip_pipeline.py -x -q -m gpu 2>&1 | tail -3
