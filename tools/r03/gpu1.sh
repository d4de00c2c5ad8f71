#!/bin/bash
# first GPU call of round 3: band-sweep kernel correctness + timing sweep + kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03a; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_roi_align_band.py -x -q > $O/pytest_band.log 2>&1; echo "pytest band rc $?" | tee -a $O/summary.txt
tail -5 $O/pytest_band.log
timeout 300 python tools/r03/band_bench.py --tag default > $O/bench_default.log 2>&1; tail -3 $O/bench_default.log | tee -a $O/summary.txt
for kv in "DTC_RA_BAND_K=16" "DTC_RA_BAND_K=12" "DTC_RA_BAND_ROWS=48" "DTC_RA_BAND_GRID=512 DTC_RA_BAND_MAXUNITS=4" "DTC_RA_BAND_GRID=2048 DTC_RA_BAND_MAXUNITS=1" "DTC_FPN_BAND_LOG2=4" "DTC_FPN_BAND_LOG2=6"; do
  env $kv timeout 300 python tools/r03/band_bench.py --tag "$kv" 2>&1 | tail -2 | tee -a $O/summary.txt
done
# kernel stats of the default configuration
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o band -- python $GRAFT_REPO_ROOT/tools/r03/band_bench.py --iters 10 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats*" | head -3
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200 | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_hip_pipeline.py tests/test_hip_roi_align.py -x -q > $O/pytest_pipe.log 2>&1; echo "pytest pipeline+roialign rc $?" | tee -a $O/summary.txt
tail -5 $O/pytest_pipe.log
