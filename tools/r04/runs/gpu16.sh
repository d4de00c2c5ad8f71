#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
for A in "--inflight 2" "--inflight 3" "--inflight 4" "--inflight 2 --eager"; do
  echo -n "bench $A | "
  timeout 600 python bench.py $A --no-cpu-baseline --sustain-seconds 0 --steps 800 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('img/s', d['value'], 'ms/step', d['ms_per_step'], 'launch', r['avg_launch_ms'])"
done
