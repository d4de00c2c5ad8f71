#!/bin/bash
# end-of-round measurement set (round 6): GPU tests with the hot-path / offscope split, smoke, kernel stats + EA counter passes per
# workload (the cfg5 legs run in contract mode: bench.py sets it), L1-fill counters of the box head, SQ / TA / TD counters of the launches
# the review asked about, the default bench line (which carries the cfg5 / cfg2 legs)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
T=${1:-r06z}
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
# SQ / TA / TD counters: cfg5 in both layouts and both modes, the C4 launch exact and fast
timeout 600 bash tools/r06/counters.sh ${T}_ctr_cfg5_nhwc_contract -- python tools/r06/ab_fused16.py --channels-last --mode contract --iters 5 > $O/ctr1.log 2>&1
timeout 600 bash tools/r06/counters.sh ${T}_ctr_cfg5_nhwc_exact -- python tools/r06/ab_fused16.py --channels-last --mode exact --iters 5 > $O/ctr2.log 2>&1
timeout 600 bash tools/r06/counters.sh ${T}_ctr_cfg5_nchw_contract -- python tools/r06/ab_fused16.py --mode contract --iters 5 > $O/ctr3.log 2>&1
timeout 600 bash tools/r06/counters.sh ${T}_ctr_cfg5_nchw_exact -- python tools/r06/ab_fused16.py --mode exact --iters 5 > $O/ctr4.log 2>&1
timeout 600 bash tools/r06/counters.sh ${T}_ctr_c4_exact -- python tools/r06/c4_launch.py --mode exact --iters 5 > $O/ctr5.log 2>&1
timeout 600 bash tools/r06/counters.sh ${T}_ctr_c4_fast -- python tools/r06/c4_launch.py --mode fast --iters 5 > $O/ctr6.log 2>&1
# committed traffic table <- the entries just collected (stamped with the hash of the kernel source that ran); bench.py below reads it
python - <<PY
import json
t = json.load(open("profiles/roialign_traffic.json"))
for w in ("cfg3", "cfg3nhwc", "cfg5", "cfg5nchw", "cfg2"):
    try:
        e = json.load(open("gpurun_out/${T}_%s/traffic_entry.json" % w))
        for k, v in e.items():
            if isinstance(v, dict): v["source"] = "profiles/r06_z_roialign_%s_pmc_raw.json" % w
        t.update(e)
    except Exception as ex:
        print("no traffic entry for", w, ex)
for lay, key in (("nchw", "cfg3_b8_nchw_f32"), ("nhwc", "cfg3_b8_nhwc_f32")):
    try:
        f = json.load(open("gpurun_out/${T}_fills_%s/l1_fills.json" % lay))
        d = t.setdefault(key + "_detail", {})
        d["l1_fill_requests"] = int(f["TCP_TCC_READ_REQ_sum"]); d["l1_fill_latency_cycles"] = round(f["l1_fill_latency_cycles"], 1)
        d["l2_read_hit_fraction"] = round(f["l2_read_hit_fraction"], 3)
        d["l1_fill_source"] = "profiles/r06_z_boxhead_l1_fill_counters_%s.json (rocprofv3 --pmc TCP_TCC_READ_REQ_sum ... -- python tools/bench_boxhead.py, tools/r04/l1_fills.sh)" % lay
    except Exception as ex:
        print("no fill counters for", lay, ex)
json.dump(t, open("profiles/roialign_traffic.json", "w"), indent=1)
json.dump(t, open("$O/roialign_traffic.json", "w"), indent=1)
PY
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
timeout 600 python bench.py --workload cfg2 --cpu-images 2 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 600 python bench.py --inflight 1 --side-steps 0 > $O/bench_inflight1.json 2> $O/bench_inflight1.err
python - <<PY | tee -a $O/summary.txt
import json
d = json.load(open("$O/bench_default.json")); r = d["roofline"]
print("default:", d["value"], "img/s", d["ms_per_step"], "ms/step | launch", r["avg_launch_ms"], "ms frac", r["frac"], "frac_harder", r.get("frac_harder"), "traffic", r["traffic"],
      "| one-stream", d["consistency"].get("one_stream_ms_per_step"), "| cpu", (d.get("cpu_baseline") or {}).get("value"), "| parity", d["cpu_baseline"]["parity_checked"]["ok"])
for k, v in d.get("other_workloads", {}).items():
    if "error" in v: print(" ", k, v); continue
    rr = v["roofline"]
    print(" ", k, v["value"], "img/s", v["ms_per_step"], "ms/step one-stream", v["one_stream_ms_per_step"], "| launch", rr["avg_launch_ms"], "frac", rr["frac"], "traffic", rr["traffic"], "| parity", v["parity_checked"]["ok"],
          "|", {x: rr[x] for x in ("exact_mode", "contract_vs_exact", "fast_mode", "bf16_output") if x in rr})
try:
    c = json.load(open("$O/bench_cfg2.json")); print("cfg2 line:", c["value"], c["ms_per_step"], c["consistency"]["one_stream_ms_per_step"], c["roofline"]["avg_launch_ms"], c["roofline"]["frac"], (c["roofline"].get("fast_mode") or {}).get("launch_ms"))
except Exception as e: print("cfg2 line failed", e)
try:
    c = json.load(open("$O/bench_inflight1.json")); print("--inflight 1:", c["value"], c["ms_per_step"])
except Exception as e: print("inflight1 failed", e)
print(open("$O/bench_default.time").read().strip().replace("\\n", " "))
PY
