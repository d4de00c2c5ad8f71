#!/bin/bash
# One entry for the round-4 experiment runs (the 28 one-off `gpurun` command files live in runs/; README.md says what each measured).
#   bash tools/r04/run.sh list          -- names + first comment line
#   bash tools/r04/run.sh <n> [args]    -- run runs/gpu<n>.sh on the GPU box (from the repo root)
cd "$(dirname "$0")" || exit 1
if [ "${1:-list}" = "list" ]; then
  for f in $(ls runs/gpu*.sh | sort -V); do printf "%-10s %s\n" "$(basename $f .sh | sed s/gpu//)" "$(grep -m1 '^#[^!]' $f | cut -c3-110)"; done
  exit 0
fi
n=$1; shift
exec bash runs/gpu$n.sh "$@"
