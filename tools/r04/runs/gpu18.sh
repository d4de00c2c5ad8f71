#!/bin/bash
# grouped 16-bit channels_last kernel (roi_align_nhwc16.hip): tests, then A/B against the one-RoI-per-workgroup kernel (DTC_RA_NHWC16=0)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_hip_roi_align.py -m gpu -x -q 2>&1 | tail -5
for rep in 1 2; do
for A in "--fp16 --channels-last" "--fp16 --channels-last --top-n 2000" "--fp16 --channels-last --mask"; do
  echo -n "one-roi $A | "; DTC_RA_NHWC16=0 timeout 300 python tools/bench_boxhead.py $A 2>&1 | tail -1
  echo -n "grouped $A | "; timeout 300 python tools/bench_boxhead.py $A 2>&1 | tail -1
done; done
for L in 0 1; do
  echo -n "NHWC16=$L bench cfg5 | "
  DTC_RA_NHWC16=$L timeout 600 python bench.py --workload cfg5 --no-cpu-baseline --sustain-seconds 0 --steps 400 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('img/s', d['value'], 'ms/step', d['ms_per_step'], 'launch', r['avg_launch_ms'], 'parity', d.get('parity_checked', {}).get('ok'))"
done
