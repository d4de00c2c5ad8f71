#!/bin/bash
# A/B of the 16-bit channels_last direct kernel's 8-channel path: old (scalar cvt + mul + add, 64-bit addresses) against
# packed (v_fma_mix_f32 + v_pk_add_f32, scalar base + 32-bit offsets).  Old library: detectorch_amd/lib/ab_old/ (built from the parent commit).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_hip_roi_align.py -m gpu -x -q 2>&1 | tail -3
OLD=$PWD/detectorch_amd/lib/ab_old/libdetectorch_hip.so
for rep in 1 2; do
for A in "--fp16 --channels-last" "--fp16 --channels-last --top-n 2000" "--fp16 --channels-last --mask"; do
  echo -n "old $A | "; DETECTORCH_HIP_LIB=$OLD timeout 300 python tools/bench_boxhead.py $A 2>&1 | tail -1
  echo -n "new $A | "; timeout 300 python tools/bench_boxhead.py $A 2>&1 | tail -1
done; done
for L in old new; do
  if [ $L = old ]; then export DETECTORCH_HIP_LIB=$OLD; else unset DETECTORCH_HIP_LIB; fi
  echo -n "$L bench cfg5 | "
  timeout 600 python bench.py --workload cfg5 --no-cpu-baseline --sustain-seconds 0 --steps 400 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('img/s', d['value'], 'ms/step', d['ms_per_step'], 'launch', r['avg_launch_ms'], 'parity', d.get('parity_checked', {}).get('ok'))"
done
