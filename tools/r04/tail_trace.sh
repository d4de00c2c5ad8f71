#!/bin/bash
# development build with the phase trace (-DDTC_PHASE_TRACE) of every short kernel -> detectorch_amd/lib/trace/libdetectorch_hip.so
cd "$(dirname "$0")/../.." || exit 1
O=detectorch_amd/lib/trace; mkdir -p $O
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wall -Wno-unused-function"
for f in detectorch_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  case $b in detections|proposals|nms|fpn|mask_paste) /opt/rocm/bin/hipcc $F -DDTC_PHASE_TRACE -c $f -o $O/$b.o & ;;
  *) cp detectorch_amd/lib/obj/$b.o $O/$b.o ;; esac
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libdetectorch_hip.so $O/*.o && echo built
