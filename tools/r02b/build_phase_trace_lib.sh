#!/bin/bash
# development build of the library with the phase trace compiled in -> detectorch_amd/lib/ptrace/libdetectorch_hip.so
cd "$(dirname "$0")/../.." || exit 1
O=detectorch_amd/lib/ptrace; mkdir -p $O
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wall -Wno-unused-function -DDTC_PHASE_TRACE"
for f in detectorch_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc $F -c $f -o $O/$b.o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libdetectorch_hip.so $O/*.o && echo built $O/libdetectorch_hip.so
