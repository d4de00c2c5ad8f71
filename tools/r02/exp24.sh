#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
for cfg in "X=1" "DTC_RA_TILE_NT=512" "DTC_RA_TILE_NT=1024" "DTC_RA_TILE_NT=512 DTC_RA_TILE_CHBLOCK=128" "DTC_RA_TILE_CHBLOCK=128" "DTC_RA_TILE_MERGE=400" "DTC_RA_TILE_NQCAP=8" "DTC_RA_TILE_NT=512 DTC_RA_TILE_MERGE=400"; do
  echo -n "$cfg : "; env $cfg timeout 300 python bench.py --workload cfg5 --no-cpu-baseline --sustain-seconds 0 --steps 200 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
