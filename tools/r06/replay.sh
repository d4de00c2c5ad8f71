#!/bin/bash
# Round 6: development builds of the library with the cluster kernel reduced to ONE of its streams
# (-DDTC_TILE_REPLAY=n, csrc/roi_align_tile.hip) -> detectorch_amd/lib/replay<n>/libdetectorch_hip.so.  Run from the repo root on the
# CPU container (hipcc cross-compiles); the .so files travel to the GPU box; tools/r06/replay_run.sh measures them.
cd "$(dirname "$0")/../.." || exit 1
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable"
for n in 1 2 3 4; do
  O=detectorch_amd/lib/replay$n; mkdir -p $O
  for f in detectorch_amd/csrc/*.hip; do
    b=$(basename $f .hip)
    if [ $b = roi_align_tile ]; then /opt/rocm/bin/hipcc $F -DDTC_TILE_REPLAY=$n -c $f -o $O/$b.o || exit 1
    else cp detectorch_amd/lib/obj/$b.o $O/$b.o; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libdetectorch_hip.so $O/*.o && rm $O/*.o && echo built $O
done
