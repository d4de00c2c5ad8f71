#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04h; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_fpn_det_mask.py tests/test_hip_pipeline.py tests/test_hip_nms.py tests/test_hip_detector.py -x -q 2>&1 | tail -6 | tee $O/tests.log
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats -- python bench.py --steps 20 --warmup 2 --eager --inflight 1 --no-cpu-baseline --sustain-seconds 0 > $O/stats.log 2>&1
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/stats_kernel_stats.csv")))
tot = 0
for r in rows[:22]:
    print("%-70s calls %5s avg %9.1f us  tot %8.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
timeout 300 python bench.py --no-cpu-baseline --sustain-seconds 0 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('default: value', d['value'], 'ms/step', d['ms_per_step'], 'launch', r['avg_launch_ms'], 'one-stream', d['consistency']['one_stream_ms_per_step'])"
