#!/bin/bash
# development build of the library with the tile kernel's phase trace compiled in -> detectorch_amd/lib/trace/libdetectorch_hip.so
cd "$(dirname "$0")/../.." || exit 1
O=detectorch_amd/lib/trace; mkdir -p $O
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wall -Wno-unused-function"
for f in detectorch_amd/csrc/*.hip; do
  b=$(basename $f .hip); X=""; [ $b = roi_align_tile ] && X="-DDTC_TILE_TRACE ${TRACE_DEFS:-}"
  /opt/rocm/bin/hipcc $F $X -c $f -o $O/$b.o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libdetectorch_hip.so $O/*.o && echo built $O/libdetectorch_hip.so
