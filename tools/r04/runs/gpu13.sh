#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_hip_fpn_det_mask.py tests/test_hip_pipeline.py tests/test_hip_detector.py -x -q 2>&1 | tail -3
for W in cfg3 cfg5; do
O=gpurun_out/r04j_$W; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats -- python bench.py --workload $W --steps 20 --warmup 2 --eager --inflight 1 --no-cpu-baseline --sustain-seconds 0 > $O/stats.log 2>&1
python - <<PY
import csv
print("== $W")
for r in list(csv.DictReader(open("$O/stats_kernel_stats.csv")))[:24]:
    if "dtc::" in r["Name"] and "roi_align" not in r["Name"]:
        print("%-60s calls %4s avg %8.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
timeout 300 python bench.py --workload $W --no-cpu-baseline --sustain-seconds 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$W: img/s', d['value'], 'ms/step', d['ms_per_step'], 'launch', r['avg_launch_ms'], 'one-stream', d['consistency']['one_stream_ms_per_step'])"
done
