#!/bin/bash
# LDS per workgroup of the cluster kernel on 16-bit NCHW maps (16-bit LDS image): sweep
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
for KB in 36 38 40 44 52; do
  export DTC_RA_TILE_LDS16_KB=$KB
  for A in "--fp16" "--fp16 --top-n 2000" "--fp16 --mask"; do
    echo -n "$KB KB $A | "; timeout 300 python tools/bench_boxhead.py $A 2>&1 | tail -1
  done
  echo -n "$KB KB bench cfg5 --nchw | "
  timeout 600 python bench.py --workload cfg5 --nchw --no-cpu-baseline --sustain-seconds 0 --steps 400 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('img/s', d['value'], 'ms/step', d['ms_per_step'], 'launch', r['avg_launch_ms'], 'harder', (r.get('harder_set') or {}).get('launch_ms'))"
done
