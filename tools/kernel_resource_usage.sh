#!/bin/bash
# Compiler view of every kernel of the library: VGPRs / SGPRs / scratch / occupancy / static LDS (hipcc -Rpass-analysis=kernel-resource-usage).
#   bash tools/kernel_resource_usage.sh > profiles/<round>_kernel_resource_usage.txt      (CPU: hipcc cross-compiles)
cd "$(dirname "$0")/.." || exit 1
echo "# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage, tree $(git rev-parse --short HEAD); name | VGPRs | SGPRs | scratch B/lane | occupancy waves/SIMD | static LDS B"
for f in detectorch_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC --cuda-device-only -c $f -o /tmp/kru.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys, re, subprocess
cur = None; rows = []
for l in sys.stdin:
    m = re.search(r'Function Name: (\S+)', l)
    if m: cur = {'n': m.group(1)}; rows.append(cur); continue
    for k, pat in (('v', r' VGPRs: (\d+)'), ('s', r'TotalSGPRs: (\d+)'), ('sc', r'ScratchSize \[bytes/lane\]: (\d+)'), ('o', r'Occupancy \[waves/SIMD\]: (\d+)'), ('l', r'LDS Size \[bytes/block\]: (\d+)')):
        m = re.search(pat, l)
        if m and cur is not None and k not in cur: cur[k] = m.group(1)
names = subprocess.run(['c++filt'], input='\n'.join(r['n'] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, name in zip(rows, names):
    name = re.sub(r'\(.*', '', name)
    print('%-92s | %4s | %4s | %3s | %2s | %6s' % (name[:92], r.get('v'), r.get('s'), r.get('sc'), r.get('o'), r.get('l')))
"
done
