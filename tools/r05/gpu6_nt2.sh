#!/bin/bash
# round 5, call 6: streaming stores in the C4 map kernel and the channels_last LDS / pipelined kernels: parity + A/B against lib/ab_old
python -m pytest tests/test_hip_roi_align.py tests/test_hip_pipeline.py -x -q -m gpu 2>&1 | tail -2
P='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("  ", d["config"]["workload_id"], d["config"]["feature_layout"], d["value"], "img/s", d["ms_per_step"], "one", d["consistency"]["one_stream_ms_per_step"], "launch", r["avg_launch_ms"], "harder", (r.get("harder_set") or {}).get("launch_ms"), "fast", (r.get("fast_mode") or {}).get("launch_ms"))'
for rep in 1 2; do
for lib in detectorch_amd/lib/ab_old/libdetectorch_hip.so detectorch_amd/lib/libdetectorch_hip.so; do
  echo "== $lib"
  for args in "--workload cfg2" "--workload cfg3 --channels-last" "--workload cfg3"; do
    DETECTORCH_HIP_LIB=$lib python bench.py $args --no-cpu-baseline --steps 200 | python -c "$P"
  done
done; done
