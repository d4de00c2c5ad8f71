#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r03b}; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_roi_align_band.py -x -q > $O/pytest_band.log 2>&1; echo "pytest band rc $?" | tee -a $O/summary.txt
tail -3 $O/pytest_band.log
timeout 300 python tools/r03/band_bench.py --tag default > $O/bench_default.log 2>&1; tail -3 $O/bench_default.log | tee -a $O/summary.txt
for sh in 1 2; do DTC_RA_BAND_SHAPE=$sh DETECTORCH_HIP_LIB=$PWD/detectorch_amd/lib/trace/libdetectorch_hip.so timeout 300 python tools/r03/band_bench.py --tag trace$sh 2>&1 | tail -14 | tee -a $O/summary.txt; done
for kv in $EXTRA_KNOBS; do
  env ${kv//,/ } timeout 300 python tools/r03/band_bench.py --tag "$kv" 2>&1 | tail -1 | tee -a $O/summary.txt
done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o band -- python $GRAFT_REPO_ROOT/tools/r03/band_bench.py --iters 10 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-160 | tee -a $O/summary.txt
