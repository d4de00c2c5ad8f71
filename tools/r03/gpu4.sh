#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r03g}; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest -m gpu rc $?" | tee -a $O/summary.txt
tail -4 $O/pytest.log | tee -a $O/summary.txt
timeout 300 python bench.py --workload cfg2 --no-cpu-baseline > $O/bench_cfg2.json 2>$O/bench_cfg2.err; python - <<PY | tee -a $O/summary.txt
import json
try:
    d = json.load(open("$O/bench_cfg2.json")); print("cfg2 prep:", d["value"], "img/s", d["ms_per_step"], "ms/step roofline", d["roofline"]["avg_launch_ms"], d["roofline"]["frac"])
except Exception as e: print("cfg2 failed", e)
PY
DTC_RA_MAP_PREP=0 timeout 300 python bench.py --workload cfg2 --no-cpu-baseline > $O/bench_cfg2_noprep.json 2>$O/bench_cfg2_noprep.err; python - <<PY | tee -a $O/summary.txt
import json
try:
    d = json.load(open("$O/bench_cfg2_noprep.json")); print("cfg2 no prep:", d["value"], "img/s", d["ms_per_step"], "ms/step roofline", d["roofline"]["avg_launch_ms"], d["roofline"]["frac"])
except Exception as e: print("cfg2 noprep failed", e)
PY
timeout 600 python bench.py > $O/bench_default.json 2>$O/bench_default.err; python - <<PY | tee -a $O/summary.txt
import json
try:
    d = json.load(open("$O/bench_default.json")); print("cfg3:", d["value"], "img/s", d["ms_per_step"], "ms/step roofline", d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], "cpu", d.get("cpu_baseline", {}).get("value"), d["consistency"])
except Exception as e: print("cfg3 failed", e)
PY
