#!/bin/bash
# TCP (vector L1) counters of the RoIAlign launch for a few kernel variants.  bash tools/tcp_probe.sh  (GPU box, repo root)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
run() {   # tag, env assignment, bench_roialign args
  OUT=gpurun_out/tcp_$1; mkdir -p $OUT
  env $2 timeout 150 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum --kernel-trace --output-format csv -d $OUT -o c -- python tools/bench_roialign.py --sort --iters 3 $3 > $OUT/c.log 2>&1 < /dev/null
  env $2 timeout 60 python tools/bench_roialign.py --sort $3 2>&1 | tail -1
  python - <<PY
import csv, collections
res = collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/c_counter_collection.csv")):
    if "roi_align" in r["Kernel_Name"]: res[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("  $1", {k: round(sum(v) / len(v) / 1e6, 2) for k, v in res.items()}, "(millions per launch)")
PY
}
run default "DTC_X=0" ""
run row4 "DTC_RA_ROW4=1" ""
run nhwc_direct "DTC_X=0" "--nhwc"
run nhwc_lds "DTC_ROIALIGN_NO_NHWC_DIRECT=1" "--nhwc"
