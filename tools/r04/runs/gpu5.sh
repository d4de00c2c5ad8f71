#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04e; mkdir -p $O
timeout 600 python bench.py --cpu-images 2 > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
timeout 600 python bench.py --channels-last --no-cpu-baseline > $O/bench_nhwc.json 2> $O/bench_nhwc.err; tail -3 $O/bench_nhwc.err
python - <<PY
import json
for n in ("default", "nhwc"):
    try:
        d = json.load(open("$O/bench_%s.json" % n)); r = d["roofline"]
        print(n, d["value"], "img/s", d["ms_per_step"], "ms/step | launch", r["avg_launch_ms"], r["launch_ms_min_median_max"], "frac", r["frac"], "| harder", r.get("harder_set"), "| one-stream", d["consistency"].get("one_stream_ms_per_step"), "| sustained", d["consistency"]["sustained"])
    except Exception as e:
        print(n, "failed", e)
PY
