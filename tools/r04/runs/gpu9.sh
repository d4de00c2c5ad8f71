#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_hip_fpn_det_mask.py tests/test_hip_pipeline.py tests/test_hip_detector.py -x -q 2>&1 | tail -3
DETECTORCH_HIP_LIB=$PWD/detectorch_amd/lib/trace/libdetectorch_hip.so python tools/r04/det_trace.py 2>&1 | tail -12
bash tools/r04/runs/gpu8.sh 2>&1 | grep -E "det_candidates|det_finalize|default:"
