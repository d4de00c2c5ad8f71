#!/bin/bash
# NCHW 16-bit maps: 16-bit LDS image in the cluster kernel (new) against the float32 image (library in detectorch_amd/lib/ab_old)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_hip_roi_align.py tests/test_hip_pipeline.py -m gpu -x -q 2>&1 | tail -2
OLD=$PWD/detectorch_amd/lib/ab_old/libdetectorch_hip.so
for rep in 1 2; do
for A in "--fp16" "--fp16 --top-n 2000" "--fp16 --mask"; do
  echo -n "old $A | "; DETECTORCH_HIP_LIB=$OLD timeout 300 python tools/bench_boxhead.py $A 2>&1 | tail -1
  echo -n "new $A | "; timeout 300 python tools/bench_boxhead.py $A 2>&1 | tail -1
  echo -n "new 38 KB $A | "; DTC_RA_TILE_LDS16_KB=38 timeout 300 python tools/bench_boxhead.py $A 2>&1 | tail -1
done; done
for L in old new new38; do
  unset DETECTORCH_HIP_LIB DTC_RA_TILE_LDS16_KB
  if [ $L = old ]; then export DETECTORCH_HIP_LIB=$OLD; fi
  if [ $L = new38 ]; then export DTC_RA_TILE_LDS16_KB=38; fi
  echo -n "$L bench cfg5 --nchw | "
  timeout 600 python bench.py --workload cfg5 --nchw --no-cpu-baseline --sustain-seconds 0 --steps 400 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('img/s', d['value'], 'ms/step', d['ms_per_step'], 'launch', r['avg_launch_ms'])"
done
