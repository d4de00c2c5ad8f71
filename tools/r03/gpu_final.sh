#!/bin/bash
# end-of-round measurement set: GPU tests, bench lines of the three workloads, kernel stats + EA counter passes per workload
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r03final}; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest -m gpu rc $?" | tee -a $O/summary.txt; tail -2 $O/pytest.log | tee -a $O/summary.txt
for w in cfg3 cfg5 cfg2; do BENCH_ARGS="--workload $w" timeout 900 bash tools/collect_profiles.sh ${1:-r03final}_$w > $O/collect_$w.log 2>&1; done
# committed traffic table <- the entries just collected (stamped with the hash of the kernel source that ran), so that the bench
# lines below report roofline.traffic; the merged file is copied out for the commit
python - <<PY
import json
t = json.load(open("profiles/roialign_traffic.json"))
for w in ("cfg3", "cfg5", "cfg2"):
    try:
        e = json.load(open("gpurun_out/${1:-r03final}_%s/traffic_entry.json" % w))
        for k, v in e.items():
            if isinstance(v, dict): v["source"] = "profiles/r03_q_roialign_%s_pmc_raw.json" % w
        t.update(e)
    except Exception as ex:
        print("no traffic entry for", w, ex)
json.dump(t, open("profiles/roialign_traffic.json", "w"), indent=1)
json.dump(t, open("$O/roialign_traffic.json", "w"), indent=1)
PY
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 900 python bench.py --workload cfg5 --cpu-images 2 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 900 python bench.py --workload cfg2 --cpu-images 2 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 300 python bench.py --inflight 1 --no-cpu-baseline > $O/bench_inflight1.json 2>/dev/null
timeout 300 python bench.py --batch 16 --no-cpu-baseline > $O/bench_batch16.json 2>/dev/null
timeout 300 python tools/bench_roialign.py --sort > $O/bench_roialign_sort.log 2>&1; tail -1 $O/bench_roialign_sort.log | tee -a $O/summary.txt
for dt in fp32 bf16; do timeout 600 python tools/bench_detector.py --batched --batch 8 --dtype $dt 2>/dev/null | tail -1 | tee -a $O/summary.txt; done
timeout 600 python tools/bench_detector.py --batched --batch 8 --dtype bf16 --channels-last 2>/dev/null | tail -1 | tee -a $O/summary.txt
python - <<PY | tee -a $O/summary.txt
import json
for n in ("default", "cfg5", "cfg2", "inflight1", "batch16"):
    try:
        d = json.load(open("$O/bench_%s.json" % n))
        print(n, d["value"], "img/s", d["ms_per_step"], "ms/step | launch", d["roofline"]["avg_launch_ms"], "ms frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "| cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(n, "failed", e)
PY
