#!/bin/bash
# end-of-round record: full GPU test-suite, smoke, the bench lines quoted in DESIGN.md / README.md, kernel stats + EA traffic
mkdir -p gpurun_out/final; O=$PWD/gpurun_out/final
timeout 600 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["config"]["workload_id"], d["config"]["images_per_gpu_per_step"], d["config"]["steps_in_flight"], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("parity_checked", {}).get("ok"), d["consistency"]["sustained"])'
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo -n "default: "; python -c "$J" < $O/bench_default.json
for args in "--inflight 1" "--inflight 3" "--batch 16" "--workload cfg5 --cpu-images 2" "--workload cfg2 --cpu-images 1"; do
  tag=$(echo $args | tr -d ' -'); echo -n "bench.py $args : "
  timeout 300 python bench.py $args --sustain-seconds 0.5 $( [[ "$args" == *workload* ]] || echo --no-cpu-baseline ) > $O/bench_$tag.json 2>/dev/null; python -c "$J" < $O/bench_$tag.json
done
echo -n "harder workload (log-uniform sides): "; timeout 200 python tools/bench_roialign.py --sort 2>&1 | tail -1
bash tools/collect_profiles.sh ${1:-r02i}
