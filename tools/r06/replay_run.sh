#!/bin/bash
# On the GPU box: the box-head launch (8 x 1000 RoIs, fp32 NCHW) of the shipped kernel and of the four single-stream replays
# (tools/r06/replay.sh), bench RoIs and the harder set, three rounds each, interleaved.
cd "$GRAFT_REPO_ROOT" || exit 1
for round in 1 2 3; do
  for n in 0 1 4 2 3; do
    if [ $n = 0 ]; then lib=detectorch_amd/lib/libdetectorch_hip.so; else lib=detectorch_amd/lib/replay$n/libdetectorch_hip.so; fi
    a=$(DETECTORCH_HIP_LIB=$PWD/$lib python tools/bench_boxhead.py --iters 30 2>/dev/null | grep -o "[0-9.]* ms/launch")
    h=$(DETECTORCH_HIP_LIB=$PWD/$lib python tools/bench_boxhead.py --iters 30 --harder 2>/dev/null | grep -o "[0-9.]* ms/launch")
    echo "round $round replay $n : bench RoIs $a ; harder set $h"
  done
done
