#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
for cfg in "X=1" "DTC_RA_NO_XCD=1" "X=2" "DTC_RA_NO_XCD=1"; do echo -n "$cfg : "; env $cfg timeout 200 python tools/bench_boxhead.py 2>&1 | tail -1; done
cd /tmp
for cfg in "X=1" "DTC_RA_NO_XCD=1"; do
  rm -rf /tmp/pm; env $cfg timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/pm -o x -- python $GRAFT_REPO_ROOT/tools/bench_boxhead.py --iters 3 > /dev/null 2>&1
  python - <<PY
import csv,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open('/tmp/pm/x_counter_collection.csv')):
    if 'roi_align_fwd_tile' in r['Kernel_Name'] and int(r['Grid_Size'])==1638400: d[r['Counter_Name']].append(float(r['Counter_Value']))
print("$cfg", {k: round(sum(v)/len(v)/1e6,2) for k,v in d.items()})
PY
done
