#!/bin/bash
# development build of the library with the pipelined channels_last kernel's phase trace compiled in
#   -> detectorch_amd/lib/trace/libdetectorch_hip.so   (select it with DETECTORCH_HIP_LIB=<that path>)
cd "$(dirname "$0")/../.." || exit 1
O=detectorch_amd/lib/trace; mkdir -p $O
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wall -Wno-unused-function"
for f in detectorch_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  if [ $b = roi_align_nhwc ]; then /opt/rocm/bin/hipcc $F -DDTC_NP_TRACE ${TRACE_DEFS:-} -c $f -o $O/$b.o &
  else cp detectorch_amd/lib/obj/$b.o $O/$b.o; fi
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libdetectorch_hip.so $O/*.o && echo built $O/libdetectorch_hip.so
