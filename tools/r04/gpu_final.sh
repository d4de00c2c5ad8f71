#!/bin/bash
# end-of-round measurement set: GPU tests, smoke, kernel stats + EA / TCP counter passes per workload, bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
T=${1:-r04final}
O=gpurun_out/$T; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest -m gpu rc $?" | tee -a $O/summary.txt; grep -E "passed|failed" $O/pytest.log | tail -2 | tee -a $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a $O/summary.txt
BENCH_ARGS="--workload cfg3" timeout 900 bash tools/collect_profiles.sh ${T}_cfg3 > $O/collect_cfg3.log 2>&1
BENCH_ARGS="--workload cfg3 --channels-last" timeout 900 bash tools/collect_profiles.sh ${T}_cfg3nhwc > $O/collect_cfg3nhwc.log 2>&1
BENCH_ARGS="--workload cfg5" timeout 900 bash tools/collect_profiles.sh ${T}_cfg5 > $O/collect_cfg5.log 2>&1
BENCH_ARGS="--workload cfg5 --nchw" timeout 900 bash tools/collect_profiles.sh ${T}_cfg5nchw > $O/collect_cfg5nchw.log 2>&1
BENCH_ARGS="--workload cfg2" timeout 900 bash tools/collect_profiles.sh ${T}_cfg2 > $O/collect_cfg2.log 2>&1
timeout 600 bash tools/r04/l1_fills.sh ${T}_fills_nchw > $O/fills_nchw.log 2>&1
timeout 600 bash tools/r04/l1_fills.sh ${T}_fills_nhwc --channels-last > $O/fills_nhwc.log 2>&1
# committed traffic table <- the entries just collected (stamped with the hash of the kernel source that ran)
python - <<PY
import json
t = json.load(open("profiles/roialign_traffic.json"))
for w, tag in (("cfg3", "cfg3"), ("cfg3nhwc", "cfg3nhwc"), ("cfg5", "cfg5"), ("cfg5nchw", "cfg5nchw"), ("cfg2", "cfg2")):
    try:
        e = json.load(open("gpurun_out/${T}_%s/traffic_entry.json" % w))
        for k, v in e.items():
            if isinstance(v, dict): v["source"] = "profiles/r04_z_roialign_%s_pmc_raw.json" % tag
        t.update(e)
    except Exception as ex:
        print("no traffic entry for", w, ex)
for lay, key in (("nchw", "cfg3_b8_nchw_f32"), ("nhwc", "cfg3_b8_nhwc_f32")):
    try:
        f = json.load(open("gpurun_out/${T}_fills_%s/l1_fills.json" % lay))
        d = t.setdefault(key + "_detail", {})
        d["l1_fill_requests"] = int(f["TCP_TCC_READ_REQ_sum"]); d["l1_fill_latency_cycles"] = round(f["l1_fill_latency_cycles"], 1)
        d["l2_read_hit_fraction"] = round(f["l2_read_hit_fraction"], 3)
        d["l1_fill_source"] = "profiles/r04_z_boxhead_l1_fill_counters_%s.json (rocprofv3 --pmc TCP_TCC_READ_REQ_sum ... -- python tools/bench_boxhead.py, tools/r04/l1_fills.sh)" % lay
    except Exception as ex:
        print("no fill counters for", lay, ex)
t["pure_load_ceilings_recorded"] = {"what": "tools/micro/l1_fill_ceiling.hip on MI355X (profiles/r04_a_l1_fill_ceiling_microbenchmark.txt), TB/s of L1 line fills of a kernel that only loads, 3 workgroups of 256 threads per CU",
                                    "l2_resident_per_xcd_slice": 21.6, "infinity_cache_resident_64MB": 7.7, "from_hbm": 6.2, "contiguous_l2_resident": 30.9}
json.dump(t, open("profiles/roialign_traffic.json", "w"), indent=1)
json.dump(t, open("$O/roialign_traffic.json", "w"), indent=1)
PY
timeout 900 bash tools/r04/runs/gpu19.sh > $O/nhwc16_counters.log 2>&1      # L1 / L2 / SQ counters of the grouped 16-bit channels_last kernel -> gpurun_out/r04n/counters.json
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 900 python bench.py --workload cfg5 --cpu-images 2 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 900 python bench.py --workload cfg5 --nchw --cpu-images 2 > $O/bench_cfg5_nchw.json 2> $O/bench_cfg5_nchw.err
timeout 900 python bench.py --workload cfg2 --cpu-images 2 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 600 python bench.py --channels-last --no-cpu-baseline > $O/bench_nhwc.json 2>/dev/null
timeout 300 python bench.py --inflight 1 --no-cpu-baseline > $O/bench_inflight1.json 2>/dev/null
timeout 300 python bench.py --batch 16 --no-cpu-baseline > $O/bench_batch16.json 2>/dev/null
python - <<PY | tee -a $O/summary.txt
import json
for n in ("default", "cfg5", "cfg5_nchw", "cfg2", "nhwc", "inflight1", "batch16"):
    try:
        d = json.load(open("$O/bench_%s.json" % n)); r = d["roofline"]
        print(n, d["value"], "img/s", d["ms_per_step"], "ms/step | launch", r["avg_launch_ms"], "ms frac", r["frac"], "traffic", r["traffic"],
              "| harder", (r.get("harder_set") or {}).get("launch_ms"), "| fast", (r.get("fast_mode") or {}).get("launch_ms"),
              "| one-stream", d["consistency"].get("one_stream_ms_per_step"), "| cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(n, "failed", e)
PY
