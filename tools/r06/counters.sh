#!/bin/bash
# round 6: SQ / TA / TD / TCP / TCC counters of the largest-grid RoIAlign launch of an arbitrary command, one counter group per pass
# (--pmc with --kernel-trace only).   bash tools/r06/counters.sh <tag> -- <command ...>   -> gpurun_out/<tag>/counters.json
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; shift; mkdir -p $O
[[ "$1" == "--" ]] && shift
i=0
for G in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD" \
         "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_WR" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
         "TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $O -o g$i -- "$@" > $O/g$i.log 2>&1 < /dev/null || tail -3 $O/g$i.log
done
python tools/r06/counters_post.py $O
