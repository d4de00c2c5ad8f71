#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02_exp12; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_fpn_det_mask.py tests/test_hip_pipeline.py tests/test_hip_detector.py -x -q -m gpu 2>&1 | tail -4
for cfg in "DTC_FPN_NO_FAST=1" "X=1"; do
  echo "== $cfg"; env $cfg timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$cfg -o s -- python bench.py --steps 50 --warmup 3 --eager --no-cpu-baseline --sustain-seconds 0 > $O/log_$cfg.txt 2>&1
  grep -E "fpn_collect|roi_align_fwd" $O/$cfg/s_kernel_stats.csv | cut -c1-140
  tail -c 300 $O/log_$cfg.txt | grep -o '"value": [0-9.]*, "unit": "images/sec"'
done
