#!/bin/bash
# one GPU call: full GPU test-suite, smoke, the default bench line, kernel stats + EA traffic of the eager single-stream bench
mkdir -p gpurun_out/val; O=$PWD/gpurun_out/val
timeout 600 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -4 $O/tests.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
timeout 200 python bench.py --no-cpu-baseline --sustain-seconds 0.5 --inflight 1 > $O/bench_inflight1.json 2>/dev/null
python -c "
import json
d=json.loads(open('$O/bench_inflight1.json').read().strip().splitlines()[-1]); print('inflight1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
bash tools/collect_profiles.sh ${1:-r02h}
