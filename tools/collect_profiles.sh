#!/bin/bash
# Run on the GPU box from the repo root:  bash tools/collect_profiles.sh <tag>
# Produces gpurun_out/<tag>/{kernel_stats.csv, traffic.json}; copy what should be judged into profiles/.
set -u
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
mkdir -p $OUT
CMD="python bench.py --steps 10 --warmup 2 --batch 8 --eager --inflight 1 --no-cpu-baseline --no-modes --sustain-seconds 0 ${BENCH_ARGS:-}"   # BENCH_ARGS="--workload cfg5" etc.
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- $CMD > $OUT/stats.log 2>&1 < /dev/null
# HBM-side traffic from the L2's fabric (EA) request counters, one counter group per run.  FETCH_SIZE itself is NOT used:
# on gfx950 its expression prices every read request at 64 B (TCC_BUBBLE reads 0) while almost all requests are 128 B
# (MI355X_MICROARCH.md, HBM section: "reports exactly 1/2"); the per-size request counters give the bytes directly:
#   read bytes  = 32*RDREQ_32B + 64*RDREQ_64B + 128*RDREQ_128B        write bytes = 64*WRREQ_64B + 32*(WRREQ - WRREQ_64B)
timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $OUT -o fetch -- $CMD > $OUT/fetch.log 2>&1 < /dev/null
timeout 200 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $OUT -o write -- $CMD > $OUT/write.log 2>&1 < /dev/null
python - <<PY
import csv, json, collections
out = "$OUT"
rows = list(csv.reader(open(out + "/stats_kernel_stats.csv")))
with open(out + "/kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(rows[0])
    for r in rows[1:]:
        w.writerow([r[0] if len(r[0]) <= 110 else r[0][:107] + "..."] + r[1:])
# RoIAlign launches of the kernel trace by (kernel, grid): the box-head and the mask-head launch share a kernel, the --stats average mixes them
by = collections.defaultdict(list)
for r in csv.DictReader(open(out + "/stats_kernel_trace.csv")):
    if "roi_align" in r["Kernel_Name"]:
        by[(r["Kernel_Name"].split("(")[0][-70:], int(r["Grid_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(out + "/launches_by_grid.txt", "w") as f:
    for (k, g), v in sorted(by.items(), key=lambda kv: -kv[0][1]):
        v = sorted(v)
        f.write("%-72s grid %8d  launches %3d  avg %8.1f us  median %8.1f  min %8.1f\n" % (k, g, len(v), sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3, v[0] / 1e3))
raw = collections.defaultdict(lambda: collections.defaultdict(list))      # grid -> counter -> values
for name in ("fetch", "write"):
    for r in csv.DictReader(open(out + "/%s_counter_collection.csv" % name)):
        if "roi_align" in r["Kernel_Name"]:
            raw[int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for g, cs in raw.items():
    a = {k: sum(v) / len(v) for k, v in cs.items()}
    rd = 32 * a.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * a.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * a.get("TCC_EA0_RDREQ_128B_sum", 0)
    w64 = a.get("TCC_EA0_WRREQ_64B_sum", 0)
    wr = 64 * w64 + 32 * (a.get("TCC_EA0_WRREQ_sum", 0) - w64)
    res[str(g)] = {"read_bytes": rd, "write_bytes": wr, "counters": a}
json.dump(res, open(out + "/traffic.json", "w"), indent=1)
# the entry for profiles/roialign_traffic.json, stamped with the hash of the kernel source the counters were collected on
import os, sys
sys.path.insert(0, "tools")
from kernel_hash import kernel_sha16
wl = "cfg3"
toks = os.environ.get("BENCH_ARGS", "").split()
for tok in toks:
    if tok in ("cfg2", "cfg3", "cfg5"): wl = tok
nhwc, f16 = ("--channels-last" in toks or (wl == "cfg5" and "--nchw" not in toks)), ("--fp16" in toks or wl == "cfg5")
if res:
    g = max(res, key=lambda k: int(k))
    key = "%s_b8_%s_%s" % (wl, "nhwc" if nhwc else "nchw", "f16" if f16 else "f32")
    entry = {key: int(res[g]["read_bytes"] + res[g]["write_bytes"]),
             key + "_detail": {"grid": int(g), "read_bytes": int(res[g]["read_bytes"]), "write_bytes": int(res[g]["write_bytes"]),
                               "kernel_sha16": kernel_sha16(wl, channels_last=nhwc), "source": "profiles/<round>_roialign_%s_pmc_raw.json" % wl}}
    json.dump(entry, open(out + "/traffic_entry.json", "w"), indent=1)
# the box-head launch is the one with the largest grid; bench.py reports its read + write bytes as roofline.traffic
if res:
    g = max(res, key=lambda k: int(k))
    print("box-head launch (grid %s): traffic = %d bytes" % (g, res[g]["read_bytes"] + res[g]["write_bytes"]))
print(json.dumps({g: {"read_GB": v["read_bytes"] / 1e9, "write_GB": v["write_bytes"] / 1e9} for g, v in res.items()}))
PY
head -12 $OUT/kernel_stats.csv | cut -c1-200
