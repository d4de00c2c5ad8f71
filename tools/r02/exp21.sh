#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
for args in "--inflight 1" "--inflight 2" "--inflight 3" "--inflight 4" "--inflight 2 --eager" "--inflight 2 --batch 4" "--inflight 4 --batch 4" "--inflight 2 --batch 16" "--inflight 2 --workload cfg5" "--inflight 2 --workload cfg2"; do
  echo -n "$args : "; timeout 300 python bench.py --no-cpu-baseline --sustain-seconds 0.5 $args 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['consistency'].get('sustained',{}).get('images_per_sec'))"
done
