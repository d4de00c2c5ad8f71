#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
for A in "" "--max-out 128" "--workload cfg5 --cpu-images 2" "--workload cfg5 --nchw --no-cpu-baseline" "--workload cfg2 --cpu-images 2"; do
  echo -n "bench $A | "
  timeout 600 python bench.py $A --sustain-seconds 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('img/s', d['value'], 'ms/step', d['ms_per_step'], 'launch', r['avg_launch_ms'], 'frac', r['frac'], 'one-stream', d['consistency']['one_stream_ms_per_step'], 'parity', (d.get('cpu_baseline') or {}).get('parity_checked', {}).get('ok'))"
done
