#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_roi_align.py -x -q 2>&1 | tail -3
for i in 1 2; do timeout 200 python tools/bench_boxhead.py 2>&1 | tail -1; done
timeout 200 python tools/bench_boxhead.py --mask 2>&1 | tail -1
timeout 200 python tools/bench_boxhead.py --fp16 2>&1 | tail -1
timeout 300 python tools/r02/trace_tile.py 2>&1 | grep -v amdgpu.ids
