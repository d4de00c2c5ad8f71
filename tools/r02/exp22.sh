#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
timeout 300 python tools/r02/c4_profile.py 2>&1 | grep -v amdgpu.ids | tail -9
echo -n "old kernel, box head sr=2: "; DTC_ROIALIGN_TILE=0 timeout 200 python tools/bench_boxhead.py 2>&1 | tail -1
timeout 900 python -m pytest tests/test_hip_roi_align.py tests/test_hip_pipeline.py -x -q 2>&1 | tail -3
timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --sustain-seconds 0 2>&1 | tail -1 | cut -c1-200
