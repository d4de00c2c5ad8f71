#!/bin/bash
# On the GPU box: A/B of the shipped library against a development build (detectorch_amd/lib/<name>/libdetectorch_hip.so): box-head
# launch on the bench RoIs and the harder set, mask head, cfg5 NCHW; three interleaved rounds.   bash tools/r06/ab_lib.sh <name>
cd "$GRAFT_REPO_ROOT" || exit 1
B=$PWD/detectorch_amd/lib/libdetectorch_hip.so; A=$PWD/detectorch_amd/lib/$1/libdetectorch_hip.so
for round in 1 2 3; do
  for lib in $B $A; do
    a=$(DETECTORCH_HIP_LIB=$lib python tools/bench_boxhead.py --iters 30 2>/dev/null | grep -o "[0-9.]* ms/launch")
    h=$(DETECTORCH_HIP_LIB=$lib python tools/bench_boxhead.py --iters 30 --harder 2>/dev/null | grep -o "[0-9.]* ms/launch")
    m=$(DETECTORCH_HIP_LIB=$lib python tools/bench_boxhead.py --iters 30 --mask 2>/dev/null | grep -o "[0-9.]* ms/launch")
    c=$(DETECTORCH_HIP_LIB=$lib python tools/bench_boxhead.py --iters 30 --fp16 --top-n 2000 2>/dev/null | grep -o "[0-9.]* ms/launch")
    echo "round $round $(basename $(dirname $lib)) : box $a ; harder $h ; mask $m ; cfg5 nchw (exact) $c"
  done
done
