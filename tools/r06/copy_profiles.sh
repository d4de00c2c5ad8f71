#!/bin/bash
# CPU container: the end-of-round measurement set (tools/r06/gpu_final.sh <tag>, merged back under gpurun_out/) -> profiles/r06_z_*
#   bash tools/r06/copy_profiles.sh r06y
cd "$(dirname "$0")/../.." || exit 1
T=${1:-r06z}; G=gpurun_out
cp $G/$T/summary.txt profiles/r06_z_final_summary.txt
cp $G/$T/bench_default.json profiles/r06_z_bench_default.json
cp $G/$T/bench_cfg2.json profiles/r06_z_bench_cfg2.json
cp $G/$T/bench_inflight1.json profiles/r06_z_bench_inflight1.json
for w in cfg3 cfg3nhwc cfg5 cfg5nchw cfg2; do
  cp $G/${T}_$w/kernel_stats.csv profiles/r06_z_bench_${w}_eager_kernel_stats.csv
  cp $G/${T}_$w/traffic.json profiles/r06_z_roialign_${w}_pmc_raw.json
done
( echo "# rocprofv3 --kernel-trace of the bench command per workload (tools/collect_profiles.sh): RoIAlign launches by kernel and grid size"
  echo "# (the box-head and the mask-head launch share a kernel: the --stats average in *_kernel_stats.csv mixes them)"
  for w in cfg3 cfg3nhwc cfg5 cfg5nchw cfg2; do sed "s/^/$w  /" $G/${T}_$w/launches_by_grid.txt; done ) > profiles/r06_z_roialign_launches_by_grid.txt
cp $G/$T/roialign_traffic.json profiles/roialign_traffic.json      # the table bench.py reads (entries stamped with the kernel-source hash)
cp $G/${T}_fills_nchw/l1_fills.json profiles/r06_z_boxhead_l1_fill_counters_nchw.json
cp $G/${T}_fills_nhwc/l1_fills.json profiles/r06_z_boxhead_l1_fill_counters_nhwc.json
python - <<PY
import json
g = "$G/$T"
c5 = {k: json.load(open("%s_ctr_cfg5_%s/counters.json" % (g, k))) for k in ("nhwc_contract", "nhwc_exact", "nchw_contract", "nchw_exact")}
json.dump(c5, open("profiles/r06_z_cfg5_counters.json", "w"), indent=1)
c4 = {k: json.load(open("%s_ctr_c4_%s/counters.json" % (g, k))) for k in ("exact", "fast")}
json.dump(c4, open("profiles/r06_z_c4_counters.json", "w"), indent=1)
PY
git status --short profiles | head -30
