cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03t; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o fb -- python tools/bench_detector.py --batched --dtype bf16 --channels-last --iters 5 > $O/fb.log 2>&1 < /dev/null
tail -2 $O/fb.log
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/fb_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms" % (tot / 1e6))
for r in rows[:45]:
    print("%-110s calls %5s avg %9.1f us total %9.1f us %5.1f%%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
