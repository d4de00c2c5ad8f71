#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04g; mkdir -p $O
BOX="python tools/bench_boxhead.py"
t() { echo -n "$1 | "; env $2 timeout 120 $BOX $3 2>&1 | tail -1; }
{
t "mask default" "DTC_X=0" "--mask"
t "mask cbmajor off" "DTC_RA_TILE_CBMAJOR=0" "--mask"
t "mask cb32" "DTC_RA_TILE_CHBLOCK=32" "--mask"
t "mask cb128" "DTC_RA_TILE_CHBLOCK=128" "--mask"
t "mask cb256" "DTC_RA_TILE_CHBLOCK=256" "--mask"
t "mask cb128 nqcap8" "DTC_RA_TILE_CHBLOCK=128 DTC_RA_TILE_NQCAP=8" "--mask"
t "mask nqcap8" "DTC_RA_TILE_NQCAP=8" "--mask"
t "mask nqcap2" "DTC_RA_TILE_NQCAP=2" "--mask"
t "mask lds 40" "DTC_RA_TILE_LDS_KB=40" "--mask"
t "box default" "DTC_X=0" ""
t "box cbmajor off" "DTC_RA_TILE_CBMAJOR=0" ""
t "box cb128" "DTC_RA_TILE_CHBLOCK=128" ""
t "cfg5 box default" "DTC_X=0" "--fp16 --top-n 2000"
t "cfg5 box cbmajor off" "DTC_RA_TILE_CBMAJOR=0" "--fp16 --top-n 2000"
t "cfg5 box cb64" "DTC_RA_TILE_CHBLOCK=64" "--fp16 --top-n 2000"
t "cfg5 box cb256" "DTC_RA_TILE_CHBLOCK=256" "--fp16 --top-n 2000"
t "cfg5 nhwc fp16 direct" "DTC_X=0" "--fp16 --top-n 2000 --channels-last"
} 2>&1 | tee $O/ab.log
