#!/bin/bash
# On the GPU box: the three-way replay (tools/r06/replay.sh) of the MASK-head launch (832 detection rows, 14 x 14 bins) and of cfg5 NCHW
cd "$GRAFT_REPO_ROOT" || exit 1
for round in 1 2; do
  for n in 0 1 4 2 3; do
    if [ $n = 0 ]; then lib=detectorch_amd/lib/libdetectorch_hip.so; else lib=detectorch_amd/lib/replay$n/libdetectorch_hip.so; fi
    m=$(DETECTORCH_HIP_LIB=$PWD/$lib python tools/bench_boxhead.py --iters 30 --mask 2>/dev/null | grep -o "[0-9.]* ms/launch")
    c=$(DETECTORCH_HIP_LIB=$PWD/$lib python tools/bench_boxhead.py --iters 30 --fp16 --top-n 2000 2>/dev/null | grep -o "[0-9.]* ms/launch")
    echo "round $round replay $n : mask head $m ; cfg5 nchw fp16 (exact) $c"
  done
done
