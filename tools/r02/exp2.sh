#!/bin/bash
# Round-2 experiment 2: where does the cluster kernel spend its time?  Phase ablation + PMC passes, old vs new kernel,
# on the bench workload's box-head launch (tools/bench_boxhead.py).
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02_exp2; mkdir -p $O
export TMPDIR=/tmp
B="python tools/bench_boxhead.py"
echo "== timings" | tee $O/times.log
for cfg in "DTC_ROIALIGN_TILE=0" "DTC_RA_TILE_CHBLOCK=64" "DTC_RA_TILE_CHBLOCK=128" "DTC_RA_TILE_CHBLOCK=64 DTC_RA_TILE_ABLATE=1" \
           "DTC_RA_TILE_CHBLOCK=64 DTC_RA_TILE_ABLATE=2" "DTC_RA_TILE_CHBLOCK=64 DTC_RA_TILE_ABLATE=3" "DTC_RA_TILE_CHBLOCK=64 DTC_RA_TILE_ABLATE=4" \
           "DTC_RA_TILE_CHBLOCK=64 DTC_RA_TILE_ABLATE=12" "DTC_RA_TILE_CHBLOCK=64 DTC_RA_TILE_ABLATE=14" "DTC_RA_TILE_CHBLOCK=64 DTC_RA_TILE_ABLATE=15" \
           "DTC_RA_TILE_CHBLOCK=64 DTC_RA_TILE_ABLATE=7" "DTC_RA_TILE_CHBLOCK=64 DTC_RA_TILE_ABLATE=6" \
           "DTC_RA_TILE_CHBLOCK=32" "DTC_RA_TILE_CHBLOCK=64 DTC_RA_TILE_K=3" "DTC_RA_TILE_CHBLOCK=64 DTC_RA_TILE_K=2" "DTC_RA_TILE_CHBLOCK=64 DTC_RA_TILE_K=1"; do
  echo -n "$cfg : " | tee -a $O/times.log; env $cfg timeout 200 $B 2>&1 | tail -1 | tee -a $O/times.log
done
pmc() {  # tag, env
  tag=$1; shift
  i=0
  while read -r GROUP; do
    [ -z "$GROUP" ] && continue
    i=$((i+1))
    env "$@" timeout 200 rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d $O/$tag -o g$i -- $B --iters 3 > $O/${tag}_g$i.log 2>&1 < /dev/null
  done <<'GROUPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_WRITE_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_WRITE_sum
GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES_EQ_64 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES
GROUPS
  python - $O/$tag <<'PY'
import csv, json, collections, glob, sys
out = sys.argv[1]
res = collections.defaultdict(list)
for f in sorted(glob.glob(out + "/g*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "roi_align" in r["Kernel_Name"]:
            res[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in res.items()}
json.dump(avg, open(out + "_pmc.json", "w"), indent=1)
print(out, json.dumps(avg))
PY
}
pmc old DTC_ROIALIGN_TILE=0
pmc tile64 DTC_RA_TILE_CHBLOCK=64
pmc tile128 DTC_RA_TILE_CHBLOCK=128
rm -rf $O/old $O/tile64 $O/tile128
