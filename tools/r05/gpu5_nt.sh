#!/bin/bash
# round 5, call 5: streaming (nt) output stores (cluster kernel + grouped channels_last kernel) against plain ones (old library in lib/ab_old)
P='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("  ", d["config"]["workload_id"], d["config"]["feature_layout"], d["value"], "img/s", d["ms_per_step"], "launch", r["avg_launch_ms"], "harder", (r.get("harder_set") or {}).get("launch_ms"))'
for rep in 1 2; do
for lib in detectorch_amd/lib/ab_old/libdetectorch_hip.so detectorch_amd/lib/libdetectorch_hip.so; do
  echo "== $lib"
  for args in "--workload cfg3" "--workload cfg3 --channels-last" "--workload cfg5" "--workload cfg5 --nchw"; do
    DETECTORCH_HIP_LIB=$lib python bench.py $args --no-cpu-baseline --steps 200 | python -c "$P"
  done
done; done
