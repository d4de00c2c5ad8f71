#!/bin/bash
# A/B of the window-staging variants of the NCHW RoIAlign kernel with the TCP counters.  bash tools/tcp_probe_quad.sh
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
run() {
  OUT=gpurun_out/tcp2_$1; mkdir -p $OUT
  env $2 timeout 150 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum --kernel-trace --output-format csv -d $OUT -o c -- python tools/bench_roialign.py --sort --iters 3 > $OUT/c.log 2>&1 < /dev/null
  env $2 timeout 60 python tools/bench_roialign.py --sort 2>&1 | tail -1
  python - <<PY
import csv, collections
res = collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/c_counter_collection.csv")):
    if "roi_align" in r["Kernel_Name"]: res[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("  $1", {k: round(sum(v) / len(v) / 1e6, 2) for k, v in res.items()}, "(millions per launch)")
PY
}
run rows_quad "DTC_RA_ROWSLOTS=1 DTC_RA_QUAD=1"
run rows_noquad "DTC_RA_ROWSLOTS=1"
run linear_quad "DTC_RA_QUAD=1"
run linear_noquad "DTC_X=0"
