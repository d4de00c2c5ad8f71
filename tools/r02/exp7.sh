#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02_exp7; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_roi_align.py tests/test_hip_pipeline.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
B="python tools/bench_boxhead.py"
echo "== timings" | tee $O/times.log
for cfg in "DTC_ROIALIGN_TILE=0" \
  "DTC_FPN_BAND_LOG2=5" "DTC_FPN_BAND_LOG2=4" "DTC_FPN_BAND_LOG2=3" \
  "DTC_FPN_BAND_LOG2=4 DTC_RA_TILE_MERGE=250" \
  "DTC_FPN_BAND_LOG2=4 DTC_RA_TILE_CHBLOCK=128" "DTC_FPN_BAND_LOG2=4 DTC_RA_TILE_CHBLOCK=32" "DTC_FPN_BAND_LOG2=4 DTC_RA_TILE_CHBLOCK=256" \
  "DTC_FPN_BAND_LOG2=4 DTC_RA_TILE_NT=512" "DTC_FPN_BAND_LOG2=3 DTC_RA_TILE_NT=512" "DTC_FPN_BAND_LOG2=4 DTC_RA_TILE_NT=512 DTC_RA_TILE_CHBLOCK=128" \
  "DTC_FPN_BAND_LOG2=4 DTC_RA_TILE_NT=512 DTC_RA_TILE_MERGE=250" "DTC_FPN_BAND_LOG2=5 DTC_RA_TILE_NT=512 DTC_RA_TILE_MERGE=250" "DTC_FPN_BAND_LOG2=4 DTC_RA_TILE_NT=1024 DTC_RA_TILE_MERGE=250" \
  "DTC_FPN_BAND_LOG2=4 DTC_RA_TILE_NQCAP=1" "DTC_FPN_BAND_LOG2=4 DTC_RA_TILE_NQCAP=4" \
  "DTC_FPN_BAND_LOG2=4 DTC_RA_TILE_ABLATE=1" "DTC_FPN_BAND_LOG2=4 DTC_RA_TILE_ABLATE=2" \
  "DTC_FPN_BAND_LOG2=4 DTC_RA_TILE_ABLATE=4" "DTC_FPN_BAND_LOG2=4 DTC_RA_TILE_ABLATE=12" \
  "DTC_FPN_BAND_LOG2=4 DTC_RA_TILE_ABLATE=14" "DTC_FPN_BAND_LOG2=4 DTC_RA_TILE_ABLATE=15"; do
  echo -n "$cfg : " | tee -a $O/times.log; env DTC_RA_TILE_CHBLOCK=64 $cfg timeout 200 $B 2>&1 | tail -1 | tee -a $O/times.log
done
for cfg in "DTC_ROIALIGN_TILE=0" "DTC_RA_TILE_NT=256" "DTC_RA_TILE_NT=512"; do
  echo -n "mask-head $cfg : " | tee -a $O/times.log; env $cfg timeout 200 $B --mask 2>&1 | tail -1 | tee -a $O/times.log
  echo -n "micro(log-uniform sides, band4) $cfg : " | tee -a $O/times.log; env $cfg DTC_FPN_BAND_LOG2=4 timeout 300 python tools/bench_roialign.py --sort 2>&1 | tail -1 | tee -a $O/times.log
done
