#!/bin/bash
# counters of the grouped 16-bit channels_last kernel (box-head launch, 8000 RoIs, fp16 maps): L1 / L2 / SQ, one group per pass
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04n; mkdir -p $O
BOX="python tools/bench_boxhead.py --iters 4 --fp16 --channels-last"
i=0
for G in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD" \
         "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
         "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $O -o g$i -- $BOX > $O/g$i.log 2>&1 < /dev/null || tail -3 $O/g$i.log
done
python - <<PY
import csv, glob, collections, json
res = collections.defaultdict(list)
for f in sorted(glob.glob("$O/g*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "roi_align_fwd_nhwc16" in r["Kernel_Name"]:
            res[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in res.items()}
json.dump(avg, open("$O/counters.json", "w"), indent=1)
for k, v in sorted(avg.items()): print("%-40s %16.0f" % (k, v))
PY
