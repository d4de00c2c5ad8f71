#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=detectorch_amd/lib/trace; mkdir -p $O
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wall -Wno-unused-function"
for f in detectorch_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  if [ $b = detections ]; then /opt/rocm/bin/hipcc $F -DDTC_PHASE_TRACE -c $f -o $O/$b.o &
  else cp detectorch_amd/lib/obj/$b.o $O/$b.o; fi
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libdetectorch_hip.so $O/*.o && echo built
