#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
for cfg in "X=1" "DTC_RA_CHBLOCK=256" "DTC_RA_CHBLOCK=512" "DTC_RA_CHBLOCK=1024" "DTC_RA_CHBLOCK=64" "DTC_ROIALIGN_LDS_KB=78" "DTC_ROIALIGN_LDS_KB=78 DTC_RA_CHBLOCK=256" "DTC_RA_PAIRS32=1" "DTC_RA_PAIRS32=1 DTC_RA_CHBLOCK=256"; do
  echo "== $cfg"; env $cfg timeout 300 python tools/r02/c4_profile.py 2>&1 | grep -v amdgpu.ids | tail -7
done
