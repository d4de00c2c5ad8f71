#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02_exp8; mkdir -p $O
export TMPDIR=/tmp
B="python tools/bench_boxhead.py"
echo "== timings" | tee $O/times.log
for cfg in "DTC_RA_TILE_MERGE=250" "DTC_RA_TILE_MERGE=400" "DTC_RA_TILE_MERGE=1000" "DTC_RA_TILE_MERGE=250 DTC_RA_TILE_NQCAP=4" "DTC_RA_TILE_MERGE=400 DTC_RA_TILE_NQCAP=4" \
  "DTC_RA_TILE_MERGE=400 DTC_RA_TILE_NQCAP=4 DTC_RA_TILE_CHBLOCK=128" "DTC_RA_TILE_MERGE=400 DTC_RA_TILE_NQCAP=3" "DTC_RA_TILE_MERGE=400 DTC_RA_TILE_K=4" \
  "DTC_RA_TILE_MERGE=400 DTC_RA_TILE_LDS_KB=40" "DTC_RA_TILE_MERGE=400 DTC_RA_TILE_LDS_KB=64" "DTC_RA_TILE_MERGE=400 DTC_FPN_BAND_LOG2=5" \
  "DTC_RA_TILE_MERGE=400 DTC_RA_TILE_NT=512" "DTC_RA_TILE_MERGE=400 DTC_RA_TILE_NT=512 DTC_RA_TILE_NQCAP=4" "DTC_RA_TILE_MERGE=400 DTC_RA_TILE_NT=512 DTC_RA_TILE_LDS_KB=52 DTC_RA_TILE_NQCAP=1"; do
  echo -n "$cfg : " | tee -a $O/times.log; env DTC_FPN_BAND_LOG2=4 DTC_RA_TILE_CHBLOCK=64 $cfg timeout 200 $B 2>&1 | tail -1 | tee -a $O/times.log
done
for cfg in "DTC_RA_TILE_MERGE=400" "DTC_RA_TILE_MERGE=400 DTC_RA_TILE_NQCAP=4" "DTC_RA_TILE_MERGE=150"; do
  echo -n "mask-head $cfg : " | tee -a $O/times.log; env DTC_FPN_BAND_LOG2=4 $cfg timeout 200 $B --mask 2>&1 | tail -1 | tee -a $O/times.log
  echo -n "micro(log-uniform sides, band4) $cfg : " | tee -a $O/times.log; env $cfg DTC_FPN_BAND_LOG2=4 timeout 300 python tools/bench_roialign.py --sort 2>&1 | tail -1 | tee -a $O/times.log
  echo -n "fp16 box $cfg : " | tee -a $O/times.log; env DTC_FPN_BAND_LOG2=4 DTC_RA_TILE_CHBLOCK=64 $cfg timeout 200 $B --fp16 2>&1 | tail -1 | tee -a $O/times.log
done
echo -n "fp16 box old : " | tee -a $O/times.log; env DTC_ROIALIGN_TILE=0 timeout 200 $B --fp16 2>&1 | tail -1 | tee -a $O/times.log
