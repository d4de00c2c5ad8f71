#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04i; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_roi_align.py -x -q -k "golden or bench or cfg5 or variants" 2>&1 | tail -2
for P in 0 768 736 704 672 640 576 512; do
  echo -n "persist $P | "
  DTC_RA_TILE_PERSIST=$P timeout 300 python bench.py --no-cpu-baseline --sustain-seconds 0 --steps 600 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('img/s', d['value'], 'ms/step', d['ms_per_step'], 'box launch', r['avg_launch_ms'], 'one-stream', d['consistency']['one_stream_ms_per_step'])"
done 2>&1 | tee $O/persist.log
