#!/bin/bash
# SQ counters of the pipelined channels_last kernel (box-head launch of the bench inputs)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04d; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -oE "\bSQ_[A-Z_0-9]+\b" | sort -u | tr '\n' ' ' > $O/sq_counter_names.txt
BOX="python tools/bench_boxhead.py --channels-last --iters 4"
i=0
for G in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD" \
         "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $O -o g$i -- $BOX > $O/g$i.log 2>&1 < /dev/null || tail -3 $O/g$i.log
done
python - <<PY
import csv, glob, collections, json
res = collections.defaultdict(list)
for f in sorted(glob.glob("$O/g*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "nhwc_pipe" in r["Kernel_Name"] and int(r["Grid_Size"]) > 100000:
            res[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in res.items()}
json.dump(avg, open("$O/sq.json", "w"), indent=1)
for k, v in sorted(avg.items()): print("%-32s %14.0f" % (k, v))
PY
