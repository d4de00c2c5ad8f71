#!/bin/bash
# end-of-round measurement set (round 5): GPU tests with the hot-path / offscope split, smoke, kernel stats + EA counter passes per
# workload, the default bench line (which carries the cfg5 / cfg2 legs)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
T=${1:-r05z}
O=gpurun_out/$T; mkdir -p $O
timeout 1500 python -m pytest tests -q -m "gpu and not offscope" > $O/pytest_hot.log 2>&1; echo "pytest -m 'gpu and not offscope' rc $? : $(grep -E 'passed|failed' $O/pytest_hot.log | tail -1)" | tee -a $O/summary.txt
timeout 900 python -m pytest tests -q -m "gpu and offscope" > $O/pytest_off.log 2>&1; echo "pytest -m 'gpu and offscope' rc $? : $(grep -E 'passed|failed' $O/pytest_off.log | tail -1)" | tee -a $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a $O/summary.txt
BENCH_ARGS="--workload cfg3" timeout 900 bash tools/collect_profiles.sh ${T}_cfg3 > $O/collect_cfg3.log 2>&1
BENCH_ARGS="--workload cfg3 --channels-last" timeout 900 bash tools/collect_profiles.sh ${T}_cfg3nhwc > $O/collect_cfg3nhwc.log 2>&1
BENCH_ARGS="--workload cfg5" timeout 900 bash tools/collect_profiles.sh ${T}_cfg5 > $O/collect_cfg5.log 2>&1
BENCH_ARGS="--workload cfg5 --nchw" timeout 900 bash tools/collect_profiles.sh ${T}_cfg5nchw > $O/collect_cfg5nchw.log 2>&1
BENCH_ARGS="--workload cfg2" timeout 900 bash tools/collect_profiles.sh ${T}_cfg2 > $O/collect_cfg2.log 2>&1
timeout 600 bash tools/r04/l1_fills.sh ${T}_fills_nchw > $O/fills_nchw.log 2>&1
timeout 600 bash tools/r04/l1_fills.sh ${T}_fills_nhwc --channels-last > $O/fills_nhwc.log 2>&1
# committed traffic table <- the entries just collected (stamped with the hash of the kernel source that ran); bench.py below reads it
python - <<PY
import json
t = json.load(open("profiles/roialign_traffic.json"))
for w in ("cfg3", "cfg3nhwc", "cfg5", "cfg5nchw", "cfg2"):
    try:
        e = json.load(open("gpurun_out/${T}_%s/traffic_entry.json" % w))
        for k, v in e.items():
            if isinstance(v, dict): v["source"] = "profiles/r05_z_roialign_%s_pmc_raw.json" % w
        t.update(e)
    except Exception as ex:
        print("no traffic entry for", w, ex)
for lay, key in (("nchw", "cfg3_b8_nchw_f32"), ("nhwc", "cfg3_b8_nhwc_f32")):
    try:
        f = json.load(open("gpurun_out/${T}_fills_%s/l1_fills.json" % lay))
        d = t.setdefault(key + "_detail", {})
        d["l1_fill_requests"] = int(f["TCP_TCC_READ_REQ_sum"]); d["l1_fill_latency_cycles"] = round(f["l1_fill_latency_cycles"], 1)
        d["l2_read_hit_fraction"] = round(f["l2_read_hit_fraction"], 3)
        d["l1_fill_source"] = "profiles/r05_z_boxhead_l1_fill_counters_%s.json (rocprofv3 --pmc TCP_TCC_READ_REQ_sum ... -- python tools/bench_boxhead.py, tools/r04/l1_fills.sh)" % lay
    except Exception as ex:
        print("no fill counters for", lay, ex)
json.dump(t, open("profiles/roialign_traffic.json", "w"), indent=1)
json.dump(t, open("$O/roialign_traffic.json", "w"), indent=1)
PY
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
timeout 600 python bench.py --workload cfg2 --cpu-images 2 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python - <<PY | tee -a $O/summary.txt
import json
d = json.load(open("$O/bench_default.json")); r = d["roofline"]
print("default:", d["value"], "img/s", d["ms_per_step"], "ms/step | launch", r["avg_launch_ms"], "ms frac", r["frac"], "traffic", r["traffic"], "| harder", (r.get("harder_set") or {}).get("launch_ms"),
      "| one-stream", d["consistency"].get("one_stream_ms_per_step"), "| cpu", (d.get("cpu_baseline") or {}).get("value"), "| parity", d["cpu_baseline"]["parity_checked"]["ok"])
for k, v in d.get("other_workloads", {}).items():
    print(" ", k, v["value"], "img/s", v["ms_per_step"], "ms/step one-stream", v["one_stream_ms_per_step"], "| launch", v["roofline"]["avg_launch_ms"], "frac", v["roofline"]["frac"], "traffic", v["roofline"]["traffic"], "| parity", v["parity_checked"]["ok"])
try:
    c = json.load(open("$O/bench_cfg2.json")); print("cfg2 line:", c["value"], c["ms_per_step"], c["consistency"]["one_stream_ms_per_step"], c["roofline"]["avg_launch_ms"], c["roofline"]["frac"], (c["roofline"].get("fast_mode") or {}).get("launch_ms"))
except Exception as e: print("cfg2 line failed", e)
print(open("$O/bench_default.time").read().strip().replace("\\n", " "))
PY
