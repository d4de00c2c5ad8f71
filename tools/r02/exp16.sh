#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
timeout 300 python tools/r02/trace_tile.py
timeout 300 python tools/r02/trace_tile.py --mask
