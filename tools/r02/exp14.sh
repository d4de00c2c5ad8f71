#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r02_exp14; mkdir -p $O
export TMPDIR=/tmp
B="python tools/bench_boxhead.py"
pmc() {
  tag=$1; shift
  i=0
  while read -r GROUP; do
    [ -z "$GROUP" ] && continue
    i=$((i+1))
    env "$@" timeout 200 rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d $O/$tag -o g$i -- $B --iters 3 > $O/${tag}_g$i.log 2>&1 < /dev/null
  done <<'GROUPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum TCP_TOTAL_WRITE_sum TCP_TCC_WRITE_REQ_LATENCY_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_WRITE_sum
GRBM_GUI_ACTIVE
GROUPS
  python - $O/$tag <<'PY'
import csv, json, collections, glob, sys
out = sys.argv[1]
res = collections.defaultdict(list)
for f in sorted(glob.glob(out + "/g*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "roi_align" in r["Kernel_Name"] and int(r["Grid_Size"]) > 1200000:
            res[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in res.items()}
json.dump(avg, open(out + "_pmc.json", "w"), indent=1)
j = avg
print(out, "fills %.1fM lat %.0f L1acc %.1fM EArd %.2fGB EAwr %.2fM tcpwr %.1fM LDSconf %.2f waitany %.2f valu %.1fM lds %.1fM salu %.1fM L2hit %.2f" % (
    j["TCP_TCC_READ_REQ_sum"]/1e6, j["TCP_TCC_READ_REQ_LATENCY_sum"]/j["TCP_TCC_READ_REQ_sum"], j["TCP_TOTAL_CACHE_ACCESSES_sum"]/1e6,
    j["TCC_EA0_RDREQ_128B_sum"]*128/1e9, j["TCC_EA0_WRREQ_64B_sum"]/1e6, j["TCP_TCC_WRITE_REQ_sum"]/1e6,
    j["SQ_LDS_BANK_CONFLICT"]/j["SQ_LDS_IDX_ACTIVE"], j["SQ_WAIT_ANY"]/j["SQ_WAVE_CYCLES"], j["SQ_INSTS_VALU"]/1e6, j["SQ_INSTS_LDS"]/1e6, j["SQ_INSTS_SALU"]/1e6,
    j["TCC_HIT_sum"]/(j["TCC_HIT_sum"]+j["TCC_MISS_sum"])))
PY
}
pmc tile X=1
rm -rf $O/tile
