#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02_exp9; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
for wl in cfg3 cfg5 cfg2; do
  timeout 600 python bench.py --workload $wl > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "bench $wl rc=$?"; tail -c 1500 $O/bench_$wl.json | head -c 1500; echo; tail -3 $O/bench_$wl.err
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_like.json 2>&1; tail -c 600 $O/bench_driver_like.json
