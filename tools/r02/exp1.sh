#!/bin/bash
# Round-2 experiment 1: correctness of the cluster-stationary RoIAlign kernel + first A/B timings.
# Usage (GPU box): bash tools/r02/exp1.sh     -> gpurun_out/r02_exp1/*
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02_exp1; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_roi_align.py tests/test_hip_pipeline.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 50 --warmup 5 > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$name" $O/bench_$name.json <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("%-28s img/s %8.1f  ms/step %.4f  box_launch_ms %.4f frac %.3f" % (sys.argv[1], j["value"], j["ms_per_step"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run old_lds DTC_ROIALIGN_TILE=0
for nt in 256 512 1024; do
  for band in 3 4 5; do
    run tile_nt${nt}_band${band} DTC_RA_TILE_NT=$nt DTC_FPN_BAND_LOG2=$band
  done
done
run tile_nt256_band4_cb64 DTC_RA_TILE_NT=256 DTC_FPN_BAND_LOG2=4 DTC_RA_TILE_CHBLOCK=64
run tile_nt256_band4_cb256 DTC_RA_TILE_NT=256 DTC_FPN_BAND_LOG2=4 DTC_RA_TILE_CHBLOCK=256
run tile_nt512_band4_cb64 DTC_RA_TILE_NT=512 DTC_FPN_BAND_LOG2=4 DTC_RA_TILE_CHBLOCK=64
run tile_nt256_band4_k4 DTC_RA_TILE_NT=256 DTC_FPN_BAND_LOG2=4 DTC_RA_TILE_K=4
run tile_nt256_band4_lds40 DTC_RA_TILE_NT=256 DTC_FPN_BAND_LOG2=4 DTC_RA_TILE_LDS_KB=40
# harder workload: log-uniform RoI sides (tools/bench_roialign.py), sorted visiting order
for cfg in "DTC_ROIALIGN_TILE=0" "DTC_RA_TILE_NT=256" "DTC_RA_TILE_NT=512" "DTC_RA_TILE_NT=1024"; do
  echo "== micro $cfg"; env $cfg timeout 300 python tools/bench_roialign.py --sort 2>&1 | tail -1
  echo "== micro maxside48 $cfg"; env $cfg timeout 300 python tools/bench_roialign.py --sort --max-side 48 2>&1 | tail -1
done
