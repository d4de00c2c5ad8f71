#!/bin/bash
# A/B inside one call: library in detectorch_amd/lib/ab_old (previous build) against the current one, 16-bit channels_last launches
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OLD=$PWD/detectorch_amd/lib/ab_old/libdetectorch_hip.so
for rep in 1 2; do
for A in "--fp16 --channels-last" "--fp16 --channels-last --top-n 2000" "--fp16 --channels-last --mask"; do
  echo -n "old $A | "; DETECTORCH_HIP_LIB=$OLD timeout 300 python tools/bench_boxhead.py $A 2>&1 | tail -1
  echo -n "new $A | "; timeout 300 python tools/bench_boxhead.py $A 2>&1 | tail -1
done; done
for L in old new; do
  if [ $L = old ]; then export DETECTORCH_HIP_LIB=$OLD; else unset DETECTORCH_HIP_LIB; fi
  echo -n "$L bench cfg5 | "
  timeout 600 python bench.py --workload cfg5 --no-cpu-baseline --sustain-seconds 0 --steps 400 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('img/s', d['value'], 'ms/step', d['ms_per_step'], 'launch', r['avg_launch_ms'])"
done
unset DETECTORCH_HIP_LIB
timeout 900 python -m pytest tests/test_hip_roi_align.py -m gpu -x -q -k "nhwc or bfloat16 or channels_last" 2>&1 | tail -2
