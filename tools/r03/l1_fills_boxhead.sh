# L1 line-fill counters of the bench's box-head RoIAlign launch alone (tools/bench_boxhead.py), one counter group per pass
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r03fills}; mkdir -p $O
i=0
for G in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $O -o g$i -- python tools/bench_boxhead.py --iters 5 > $O/g$i.log 2>&1 < /dev/null
done
python - <<PY
import csv, json, collections, glob
res = collections.defaultdict(list)
for f in sorted(glob.glob("$O/g*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "roi_align_fwd_tile" in r["Kernel_Name"] and int(r["Grid_Size"]) == 1638400:
            res[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in res.items()}
avg["launches_averaged"] = {k: len(v) for k, v in res.items()}
json.dump(avg, open("$O/l1_fills.json", "w"), indent=1)
print(json.dumps(avg))
PY
tail -2 $O/g1.log
