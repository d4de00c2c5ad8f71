#!/bin/bash
# Round 4, measurement set 1 (VERDICT r03 "Next round" 1a): the L2-resident L1-fill ceiling, and TCC / EA / TCP counters of
#   (i) the shipped NCHW box-head launch, (ii) the 1-image x 8000-RoI control, (iii) roi_align_fwd_nhwc_lds fp32 on the bench inputs,
# plus the dispatch-order variants of the cluster kernel (channel-block-major XCD walk, clock-phased passes) WITH their L2 counters.
#   bash tools/r04/runs/gpu1.sh   (GPU box, repo root)  -> gpurun_out/r04a/
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04a; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/l1fill tools/micro/l1_fill_ceiling.hip && timeout 120 /tmp/l1fill > $O/l1_fill_ceiling.txt 2>&1
tail -12 $O/l1_fill_ceiling.txt
PA="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
PB="TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
PC="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum"
run() {   # tag, env, passes, command...
  local tag=$1 envs=$2 passes=$3; shift 3
  env $envs timeout 120 "$@" > $O/$tag.time.log 2>&1 < /dev/null
  tail -1 $O/$tag.time.log
  for P in $passes; do
    local G; case $P in A) G=$PA;; B) G=$PB;; C) G=$PC;; esac
    env $envs timeout 200 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $O -o ${tag}_$P -- "$@" --iters 4 > $O/${tag}_$P.log 2>&1 < /dev/null
  done
}
BOX="python tools/bench_boxhead.py"
run shipped "DTC_X=0" "A B C" $BOX
run nhwc_f32 "DTC_X=0" "A B C" $BOX --channels-last
run one_image "DTC_X=0" "A B C" python tools/bench_roialign.py --sort --batch 1 --rois 8000
run eight_image "DTC_X=0" "A" python tools/bench_roialign.py --sort --batch 8 --rois 1000
run cbmajor "DTC_RA_TILE_CBMAJOR=1" "A" $BOX
run phase4 "DTC_RA_TILE_PHASE_T=400" "A" $BOX
run cbmajor_phase2 "DTC_RA_TILE_CBMAJOR=1 DTC_RA_TILE_PHASE_T=200" "A" $BOX
run cbmajor_phase4 "DTC_RA_TILE_CBMAJOR=1 DTC_RA_TILE_PHASE_T=400" "A C" $BOX
run cbmajor_phase8 "DTC_RA_TILE_CBMAJOR=1 DTC_RA_TILE_PHASE_T=800" "A" $BOX
run cbmajor_phase4_cb128 "DTC_RA_TILE_CBMAJOR=1 DTC_RA_TILE_PHASE_T=400 DTC_RA_TILE_CHBLOCK=128" "A" $BOX
run cbmajor_phase4_cb32 "DTC_RA_TILE_CBMAJOR=1 DTC_RA_TILE_PHASE_T=400 DTC_RA_TILE_CHBLOCK=32" "A" $BOX
python - <<PY
import csv, json, collections, glob, os, re
O = "$O"
res = {}
for f in sorted(glob.glob(O + "/*_counter_collection.csv")):
    tag = os.path.basename(f)[:-len("_counter_collection.csv")]
    tag = tag[:-2]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "roi_align" in r["Kernel_Name"]:
            kn = re.sub(r"<.*", "", r["Kernel_Name"].split("(")[0]).split("::")[-1]
            acc[(kn, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not acc: continue
    key = max(acc, key=lambda k: k[1])       # the box-head launch: the largest grid
    d = res.setdefault(tag, {"kernel": key[0], "grid": key[1]})
    for c, v in acc[key].items():
        d[c] = sum(v[1:]) / max(1, len(v) - 1) if len(v) > 1 else v[0]     # skip the first (cold) launch
for tag, d in res.items():
    g = lambda k: d.get(k, 0.0)
    if "TCC_HIT_sum" in d: d["l2_hit_frac"] = g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    if "TCC_EA0_RDREQ_128B_sum" in d:
        d["ea_read_bytes"] = 32 * g("TCC_EA0_RDREQ_32B_sum") + 64 * g("TCC_EA0_RDREQ_64B_sum") + 128 * g("TCC_EA0_RDREQ_128B_sum")
        d["ea_write_bytes"] = 64 * g("TCC_EA0_WRREQ_64B_sum") + 32 * (g("TCC_EA0_WRREQ_sum") - g("TCC_EA0_WRREQ_64B_sum"))
    elif "TCC_EA0_RDREQ_sum" in d:
        d["ea_read_bytes_upper"] = 128 * g("TCC_EA0_RDREQ_sum") - 96 * g("TCC_EA0_RDREQ_32B_sum")    # every non-32-B request priced at 128 B
    if "TCP_TCC_READ_REQ_sum" in d:
        d["l1_fill_bytes"] = 128 * g("TCP_TCC_READ_REQ_sum")
        d["l1_fill_latency_cycles"] = g("TCP_TCC_READ_REQ_LATENCY_sum") / max(1.0, g("TCP_TCC_READ_REQ_sum"))
    try: d["time_line"] = open(O + "/%s.time.log" % tag).read().strip().splitlines()[-1]
    except Exception: pass
json.dump(res, open(O + "/summary.json", "w"), indent=1)
for tag, d in res.items():
    print(tag, {k: (round(v, 3) if isinstance(v, float) and v < 10 else (round(v / 1e6, 2) if isinstance(v, float) else v)) for k, v in d.items() if not k.endswith("_sum")})
PY
