#!/bin/bash
# map-stationary (C4) kernel: odd row pitch of the LDS image (default) against pitch = W (DTC_RA_MAP_PITCH=0)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_hip_roi_align.py tests/test_hip_pipeline.py -m gpu -x -q -k "map or c4 or variants or cfg2 or C4" 2>&1 | tail -2
for rep in 1 2; do
for L in 0 1; do
  echo -n "MAP_PITCH=$L bench cfg2 | "
  DTC_RA_MAP_PITCH=$L timeout 600 python bench.py --workload cfg2 --no-cpu-baseline --sustain-seconds 0 --steps 200 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('img/s', d['value'], 'ms/step', d['ms_per_step'], 'launch', r['avg_launch_ms'], 'fast', (r.get('fast_mode') or {}).get('launch_ms'))"
done; done
