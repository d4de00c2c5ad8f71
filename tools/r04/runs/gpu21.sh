#!/bin/bash
# float32 channels_last maps through the grouped direct kernel (DTC_RA_NHWC_DIRECT32=1) against the LDS-DMA kernels (default);
# 16-bit maps on the harder RoI set, grouped against one RoI per workgroup
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_hip_roi_align.py -m gpu -x -q -k "nhwc or bfloat16 or channels_last" 2>&1 | tail -2
for rep in 1 2; do
for A in "--channels-last" "--channels-last --mask"; do
  echo -n "lds-dma $A | "; timeout 300 python tools/bench_boxhead.py $A 2>&1 | tail -1
  echo -n "direct  $A | "; DTC_RA_NHWC_DIRECT32=1 timeout 300 python tools/bench_boxhead.py $A 2>&1 | tail -1
done
echo -n "harder set fp32 nhwc lds-dma | "; timeout 300 python tools/bench_roialign.py --sort --nhwc 2>&1 | tail -1
echo -n "harder set fp32 nhwc direct  | "; DTC_RA_NHWC_DIRECT32=1 timeout 300 python tools/bench_roialign.py --sort --nhwc 2>&1 | tail -1
echo -n "harder set fp16 nhwc one-roi | "; DTC_RA_NHWC16=0 timeout 300 python tools/bench_roialign.py --sort --nhwc --half 2>&1 | tail -1
echo -n "harder set fp16 nhwc grouped | "; timeout 300 python tools/bench_roialign.py --sort --nhwc --half 2>&1 | tail -1
done
for L in 0 1; do
  echo -n "DIRECT32=$L bench --channels-last | "
  DTC_RA_NHWC_DIRECT32=$L timeout 600 python bench.py --channels-last --no-cpu-baseline --sustain-seconds 0 --steps 400 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('img/s', d['value'], 'ms/step', d['ms_per_step'], 'launch', r['avg_launch_ms'], 'harder', (r.get('harder_set') or {}).get('launch_ms'))"
done
