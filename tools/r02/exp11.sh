#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
B="python tools/bench_boxhead.py"
for cfg in "DTC_FPN_GROUP_LOG2=0" "DTC_FPN_GROUP_LOG2=1 DTC_FPN_CELL_LOG2=6" "DTC_FPN_GROUP_LOG2=1 DTC_FPN_CELL_LOG2=7" "DTC_FPN_GROUP_LOG2=2 DTC_FPN_CELL_LOG2=7" "DTC_FPN_GROUP_LOG2=1 DTC_FPN_CELL_LOG2=8"; do
  echo -n "$cfg : "; env $cfg timeout 200 $B 2>&1 | tail -1
done
