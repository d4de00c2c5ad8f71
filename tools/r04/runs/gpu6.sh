#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04f; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 | tee $O/tests.log
