#!/bin/bash
# CPU container: development build of the library with the cluster kernel's phase trace compiled in (-DDTC_TILE_TRACE)
#   -> detectorch_amd/lib/trace/libdetectorch_hip.so ; on the GPU box: python tools/r06/tile_trace.py [--mask|--harder|...]
cd "$(dirname "$0")/../.." || exit 1
O=detectorch_amd/lib/trace; mkdir -p $O
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable"
for f in detectorch_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  if [ $b = roi_align_tile ]; then /opt/rocm/bin/hipcc $F -DDTC_TILE_TRACE -c $f -o $O/$b.o || exit 1
  else cp detectorch_amd/lib/obj/$b.o $O/$b.o; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libdetectorch_hip.so $O/*.o && echo built $O/libdetectorch_hip.so
