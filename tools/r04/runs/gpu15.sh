#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_hip_roi_align.py -x -q -k "nhwc or bfloat16 or variants or cfg5" 2>&1 | tail -3
BOX="python tools/bench_boxhead.py"
echo -n "cfg5 nhwc fp16 box | "; timeout 120 $BOX --channels-last --fp16 --top-n 2000 2>&1 | tail -1
echo -n "cfg3 nhwc fp16 box | "; timeout 120 $BOX --channels-last --fp16 2>&1 | tail -1
echo -n "nhwc fp16 mask | "; timeout 120 $BOX --channels-last --fp16 --mask 2>&1 | tail -1
timeout 300 python bench.py --workload cfg5 --no-cpu-baseline --sustain-seconds 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('cfg5: img/s', d['value'], 'ms/step', d['ms_per_step'], 'launch', r['avg_launch_ms'], 'frac', r['frac'], 'one-stream', d['consistency']['one_stream_ms_per_step'])"
