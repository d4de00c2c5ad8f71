#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r03f}; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_roi_align_band.py tests/test_hip_pipeline.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt
tail -3 $O/pytest.log
timeout 300 python tools/r03/band_bench.py --mask --tag default 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $O/summary.txt
for kv in $EXTRA_KNOBS; do
  env ${kv//,/ } timeout 300 python tools/r03/band_bench.py --tag "$kv" 2>&1 | tail -1 | tee -a $O/summary.txt
done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o band -- python $GRAFT_REPO_ROOT/tools/r03/band_bench.py --iters 10 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep "dtc::" "$f" | cut -c1-170 | tee -a $O/summary.txt
