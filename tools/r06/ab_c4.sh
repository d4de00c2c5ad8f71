#!/bin/bash
# On the GPU box: the C4 (cfg2) RoIAlign launch, exact and fast mode, for several builds side by side (detectorch_amd/lib/<name>/; "lib" = shipped)
#   bash tools/r06/ab_c4.sh <name> <name> ...
cd "$GRAFT_REPO_ROOT" || exit 1
for round in 1 2 3; do
  for n in "$@"; do
    if [ $n = lib ]; then lib=$PWD/detectorch_amd/lib/libdetectorch_hip.so; else lib=$PWD/detectorch_amd/lib/$n/libdetectorch_hip.so; fi
    e=$(DETECTORCH_HIP_LIB=$lib python tools/r06/c4_launch.py --mode exact --iters 30 2>/dev/null | grep -o "[0-9.]* ms per launch")
    f=$(DETECTORCH_HIP_LIB=$lib python tools/r06/c4_launch.py --mode fast --iters 30 2>/dev/null | grep -o "[0-9.]* ms per launch")
    echo "round $round $n : exact $e ; fast $f"
  done
done
