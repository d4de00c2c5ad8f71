#!/bin/bash
# On the GPU box: several development builds (detectorch_amd/lib/<name>/libdetectorch_hip.so) side by side, interleaved rounds.
#   bash tools/r06/ab_libs.sh <name> <name> ...      ("lib" = the shipped library)
cd "$GRAFT_REPO_ROOT" || exit 1
for round in 1 2 3; do
  for n in "$@"; do
    if [ $n = lib ]; then lib=$PWD/detectorch_amd/lib/libdetectorch_hip.so; else lib=$PWD/detectorch_amd/lib/$n/libdetectorch_hip.so; fi
    a=$(DETECTORCH_HIP_LIB=$lib python tools/bench_boxhead.py --iters 30 2>/dev/null | grep -o "[0-9.]* ms/launch")
    h=$(DETECTORCH_HIP_LIB=$lib python tools/bench_boxhead.py --iters 30 --harder 2>/dev/null | grep -o "[0-9.]* ms/launch")
    m=$(DETECTORCH_HIP_LIB=$lib python tools/bench_boxhead.py --iters 30 --mask 2>/dev/null | grep -o "[0-9.]* ms/launch")
    c=$(DETECTORCH_HIP_LIB=$lib python tools/bench_boxhead.py --iters 30 --fp16 --top-n 2000 2>/dev/null | grep -o "[0-9.]* ms/launch")
    echo "round $round $n : box $a ; harder $h ; mask $m ; cfg5 nchw (exact) $c"
  done
done
