#!/bin/bash
# Memory-system PMC passes over the box-head RoIAlign launch alone (tools/bench_roialign.py --sort, 8000 RoIs).
# Run on the GPU box from the repo root:  bash tools/collect_memsys.sh <tag> [extra bench_roialign args]
# One counter group per run (separate --pmc passes), kernel-trace only.  Output: gpurun_out/<tag>/memsys.json
set -u
TAG=${1:-memsys}; shift || true
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
mkdir -p $OUT
CMD="python tools/bench_roialign.py --sort --iters 5 $*"
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d $OUT -o g$i -- $CMD > $OUT/g$i.log 2>&1 < /dev/null
done <<'GROUPS'
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_BUBBLE_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
GROUPS
python - <<PY
import csv, json, collections, glob
out = "$OUT"
res = collections.defaultdict(list)
for f in sorted(glob.glob(out + "/g*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "roi_align" in r["Kernel_Name"]:
            res[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in res.items()}
json.dump(avg, open(out + "/memsys.json", "w"), indent=1)
print(json.dumps(avg, indent=1))
PY
