#!/bin/bash
# does the LDS footprint of a short kernel (nms_reduce_lds: 136 KB = a whole CU) keep it from running beside the other stream's RoIAlign?
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2 3; do
for L in lds nolds; do
  if [ $L = nolds ]; then export DTC_NMS_NO_LDS_WALK=1; else unset DTC_NMS_NO_LDS_WALK; fi
  echo -n "reduce $L | "
  timeout 600 python bench.py --no-cpu-baseline --sustain-seconds 0 --steps 800 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('img/s', d['value'], 'ms/step', d['ms_per_step'], 'launch', r['avg_launch_ms'], 'one-stream', d['consistency'].get('one_stream_ms_per_step'))"
done; done
