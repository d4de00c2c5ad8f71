cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for bl in 3 4 5 6; do
  export DTC_FPN_BAND_LOG2=$bl
  mkdir -p gpurun_out/tile$bl
  timeout 120 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('band_log2', $bl, d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d gpurun_out/tile$bl -o fetch -- python bench.py --steps 3 --warmup 1 --batch 8 --eager --no-cpu-baseline > gpurun_out/tile$bl/fetch.log 2>&1 < /dev/null
  python - <<PY
import csv,collections
per=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("gpurun_out/tile$bl/fetch_counter_collection.csv")):
    if "roi_align" in r["Kernel_Name"]: per[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g,c in per.items():
    a={k:sum(v)/len(v) for k,v in c.items()}
    print("   grid",g, round((64*a["TCC_EA0_RDREQ_64B_sum"]+128*a["TCC_EA0_RDREQ_128B_sum"])/1e9,3), "GB read")
PY
done
