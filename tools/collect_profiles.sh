#!/bin/bash
# Run on the GPU box from the repo root:  bash tools/collect_profiles.sh <tag>
# Produces gpurun_out/<tag>/{kernel_stats.csv, traffic.json}; copy what should be judged into profiles/.
set -u
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
mkdir -p $OUT
CMD="python bench.py --steps 5 --warmup 2 --batch 8 --eager --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- $CMD > $OUT/stats.log 2>&1 < /dev/null
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o fetch -- $CMD > $OUT/fetch.log 2>&1 < /dev/null
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT -o write -- $CMD > $OUT/write.log 2>&1 < /dev/null
python - <<PY
import csv, json, collections
out = "$OUT"
rows = list(csv.reader(open(out + "/stats_kernel_stats.csv")))
with open(out + "/kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(rows[0])
    for r in rows[1:]:
        w.writerow([r[0] if len(r[0]) <= 110 else r[0][:107] + "..."] + r[1:])
res = {}
for name in ("fetch", "write"):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(out + "/%s_counter_collection.csv" % name)):
        if "roi_align" in r["Kernel_Name"]:
            per[int(r["Grid_Size"])].append(float(r["Counter_Value"]))
    res[name] = {str(g): sum(v) / len(v) for g, v in per.items()}
json.dump(res, open(out + "/traffic.json", "w"), indent=1)
print(json.dumps(res))
PY
head -12 $OUT/kernel_stats.csv | cut -c1-200
