#!/bin/bash
# full GPU test-suite, phase trace, kernel stats + bench lines
mkdir -p gpurun_out/all; O=$PWD/gpurun_out/all
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -4 $O/tests.log
python tools/r02b/phase_trace.py 2>&1 | grep -v amdgpu.ids | tee $O/phase_trace.txt
SKIP_TESTS=1 bash tools/r02b/tail_stats.sh
