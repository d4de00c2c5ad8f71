#!/bin/bash
# pre-NMS ranking at K = 6000 (cfg2): rank by counting across workgroups (default) against the one-workgroup bitonic sort (DTC_RPN_RANK=0)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_hip_proposals.py tests/test_hip_pipeline.py tests/test_hip_detector.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do
for L in 0 1; do
  echo -n "RPN_RANK=$L bench cfg2 | "
  DTC_RPN_RANK=$L timeout 600 python bench.py --workload cfg2 --no-cpu-baseline --sustain-seconds 0 --steps 200 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('img/s', d['value'], 'ms/step', d['ms_per_step'], 'launch', r['avg_launch_ms'], 'one-stream', d['consistency'].get('one_stream_ms_per_step'))"
done; done
O=gpurun_out/r04rank; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats -- python bench.py --workload cfg2 --steps 10 --warmup 2 --batch 8 --eager --inflight 1 --no-cpu-baseline --sustain-seconds 0 > $O/stats.log 2>&1 < /dev/null
grep -E "rpn_|nms_" $O/stats_kernel_stats.csv | cut -c1-140
