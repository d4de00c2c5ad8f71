#!/bin/bash
# the numbers quoted in DESIGN.md / README.md for the end of round 2 (one MI355X)
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
J='import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload_id"], d["config"]["images_per_gpu_per_step"], d["config"]["steps_in_flight"], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], (d.get("cpu_baseline") or {}).get("parity_checked", {}).get("ok"))'
for args in "" "--inflight 1" "--workload cfg5" "--workload cfg5 --inflight 1" "--workload cfg2" "--workload cfg2 --inflight 1" "--batch 16 --no-cpu-baseline" "--batch 32 --no-cpu-baseline"; do
  echo -n "bench.py $args : "; timeout 400 python bench.py $args 2>&1 | tail -1 | python -c "$J"
done
echo -n "harder workload (log-uniform sides): "; timeout 200 python tools/bench_roialign.py --sort 2>&1 | tail -1
echo -n "mask head: "; timeout 200 python tools/bench_boxhead.py --mask 2>&1 | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
