#!/bin/bash
# does the LDS the cluster kernel leaves free on a CU (3 workgroups x 52 KB = 156 of 160 KB) matter for the other stream's short kernels?
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for KB in 52 48 44 40; do
  echo -n "LDS $KB KB | "
  DTC_RA_TILE_LDS_KB=$KB timeout 600 python bench.py --no-cpu-baseline --sustain-seconds 0 --steps 800 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('img/s', d['value'], 'ms/step', d['ms_per_step'], 'launch', r['avg_launch_ms'], 'harder', (r.get('harder_set') or {}).get('launch_ms'), 'one-stream', d['consistency'].get('one_stream_ms_per_step'))"
done; done
