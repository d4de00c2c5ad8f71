#!/bin/bash
# Round 4, set 2: the pipelined channels_last RoIAlign kernel -- parity tests, then A/B against the round-3 kernel and an LDS sweep.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04b; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_roi_align.py -x -q -k "nhwc" 2>&1 | tail -15 | tee $O/tests.log
BOX="python tools/bench_boxhead.py"
t() { echo -n "$1 | "; env $2 timeout 120 $BOX --channels-last $3 2>&1 | tail -1; }
{
t "old kernel 40 KB" "DTC_RA_NHWC_PIPE=0" ""
t "pipe default (53 KB, 3 WG/CU)" "DTC_X=0" ""
t "pipe 40 KB 4 WG" "DTC_RA_NHWC_LDS_KB=40" ""
t "pipe 46 KB 3 WG" "DTC_RA_NHWC_LDS_KB=46" ""
t "pipe 64 KB 2 WG" "DTC_RA_NHWC_LDS_KB=64" ""
t "pipe 78 KB 2 WG" "DTC_RA_NHWC_LDS_KB=78" ""
t "pipe 104 KB 1 WG" "DTC_RA_NHWC_LDS_KB=104" ""
t "pipe 53 KB, 2 WG/CU launched" "DTC_RA_NHWC_WGS=2" ""
t "pipe 53 KB, 4 launched (oversubscribed)" "DTC_RA_NHWC_WGS=4" ""
t "pipe 53 KB, no XCD slices" "DTC_RA_NO_XCD=1" ""
t "fp16 direct (shipped)" "DTC_X=0" "--fp16"
t "fp16 pipe 78 KB" "DTC_RA_NHWC_PIPE16=1" "--fp16"
t "fp16 pipe 53 KB" "DTC_RA_NHWC_PIPE16=1 DTC_RA_NHWC_LDS_KB=53" "--fp16"
t "fp16 pipe 104 KB" "DTC_RA_NHWC_PIPE16=1 DTC_RA_NHWC_LDS_KB=104" "--fp16"
t "fp16 cfg5 (2000 rois) direct" "DTC_X=0" "--fp16 --top-n 2000"
t "fp16 cfg5 (2000 rois) pipe 78" "DTC_RA_NHWC_PIPE16=1" "--fp16 --top-n 2000"
t "mask fp32 old path" "DTC_RA_NHWC_PIPE=0" "--mask"
t "mask fp32 pipe" "DTC_X=0" "--mask"
t "mask fp16 direct" "DTC_X=0" "--mask --fp16"
t "mask fp16 pipe" "DTC_RA_NHWC_PIPE16=1" "--mask --fp16"
echo -n "NCHW shipped (same box) | "; timeout 120 $BOX 2>&1 | tail -1
} 2>&1 | tee $O/ab.log
