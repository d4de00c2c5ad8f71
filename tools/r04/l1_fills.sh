#!/bin/bash
# L1 line-fill / L2 counters of the box-head RoIAlign launch (tools/bench_boxhead.py $2...), one counter group per pass -> $O/l1_fills.json
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/$1; shift; mkdir -p $O
i=0
for G in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $O -o g$i -- python tools/bench_boxhead.py --iters 5 "$@" > $O/g$i.log 2>&1 < /dev/null
done
python - <<PY
import csv, json, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$O/g*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "roi_align" in r["Kernel_Name"]:
            acc[int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
g = max(acc)
avg = {k: sum(v[1:]) / max(1, len(v) - 1) for k, v in acc[g].items()}
avg["grid"] = g
avg["l1_fill_latency_cycles"] = avg["TCP_TCC_READ_REQ_LATENCY_sum"] / max(1.0, avg["TCP_TCC_READ_REQ_sum"])
avg["l2_read_hit_fraction"] = avg["TCC_HIT_sum"] / max(1.0, avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"])
json.dump(avg, open("$O/l1_fills.json", "w"), indent=1)
print(json.dumps(avg))
PY
