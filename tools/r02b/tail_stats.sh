#!/bin/bash
# kernel durations of one eager single-stream bench run (the small-kernel tail), + the tests that cover the touched kernels
mkdir -p gpurun_out/tail; O=$PWD/gpurun_out/tail
if [ "${SKIP_TESTS:-0}" != 1 ]; then
timeout 900 python -m pytest tests/test_hip_fpn_det_mask.py tests/test_hip_pipeline.py tests/test_hip_detector.py tests/test_hip_proposals.py tests/test_hip_nms.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -4 $O/tests.log
fi
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --eager --inflight 1 --no-cpu-baseline --sustain-seconds 0 > $O/bench_eager.json 2> $O/prof.err
f=$(find $O/prof -name '*kernel_stats.csv' | head -1)
cp $f $O/kernel_stats.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/kernel_stats.csv")))
tot = 0
for r in rows:
    n = r["Name"]
    if "dtc::" not in n: continue
    short = n.split("dtc::")[1].split("(")[0][:40]
    print("%-42s calls %4s avg %8.1f us  min %8.1f" % (short, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
cd $GRAFT_REPO_ROOT
timeout 200 python bench.py --no-cpu-baseline --sustain-seconds 0.5 > $O/bench_default.json 2>/dev/null
timeout 200 python bench.py --no-cpu-baseline --sustain-seconds 0.5 --inflight 1 > $O/bench_inflight1.json 2>/dev/null
python - <<PY
import json
for t in ("default", "inflight1"):
    try:
        d = json.loads(open("$O/bench_%s.json" % t).read().strip().splitlines()[-1]); print(t, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"])
    except Exception as e: print(t, "FAILED", e)
PY
