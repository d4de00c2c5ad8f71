#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_roi_align.py tests/test_hip_pipeline.py -x -q 2>&1 | tail -2
for i in 1 2; do timeout 200 python tools/bench_boxhead.py 2>&1 | tail -1; timeout 200 python tools/bench_boxhead.py --mask 2>&1 | tail -1; done
timeout 200 python tools/r02/trace_tile.py 2>&1 | grep -v amdgpu | sed -n 2,16p
timeout 300 python bench.py --no-cpu-baseline --sustain-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
