#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04c; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_roi_align.py -x -q -k "nhwc" 2>&1 | tail -8 | tee $O/tests.log
BOX="python tools/bench_boxhead.py"
t() { echo -n "$1 | "; env $2 timeout 120 $BOX --channels-last $3 2>&1 | tail -1; }
{
t "old kernel 40 KB" "DTC_RA_NHWC_PIPE=0" ""
t "pipe v2 default (78 KB, 2 WG/CU)" "DTC_X=0" ""
t "pipe v2 64 KB" "DTC_RA_NHWC_LDS_KB=64" ""
t "pipe 70 KB" "DTC_RA_NHWC_LDS_KB=70" ""
t "pipe 53 KB 3 WG" "DTC_RA_NHWC_LDS_KB=53" ""
t "pipe v2 104 KB 1 WG" "DTC_RA_NHWC_LDS_KB=104" ""
t "pipe v2 156 KB 1 WG" "DTC_RA_NHWC_LDS_KB=156" ""
t "fp16 direct (shipped)" "DTC_X=0" "--fp16"
t "fp16 pipe 78 KB" "DTC_RA_NHWC_PIPE16=1" "--fp16"
t "fp16 cfg5 direct" "DTC_X=0" "--fp16 --top-n 2000"
t "fp16 cfg5 pipe 78" "DTC_RA_NHWC_PIPE16=1" "--fp16 --top-n 2000"
t "mask fp32 old path" "DTC_RA_NHWC_PIPE=0" "--mask"
t "mask fp32 pipe" "DTC_X=0" "--mask"
t "mask fp16 direct" "DTC_X=0" "--mask --fp16"
t "mask fp16 pipe" "DTC_RA_NHWC_PIPE16=1" "--mask --fp16"
echo -n "NCHW shipped (same box) | "; timeout 120 $BOX 2>&1 | tail -1
} 2>&1 | tee $O/ab.log
export DETECTORCH_HIP_LIB=$PWD/detectorch_amd/lib/trace/libdetectorch_hip.so
for kb in 78; do
  echo "== LDS $kb KB"; DTC_RA_NHWC_LDS_KB=$kb timeout 120 python tools/r04/np_trace.py 2>&1 | tail -14
done | tee $O/trace.log
