#!/bin/bash
# where does the two-streams-in-flight schedule lose time?  kernel trace of the default bench, then: share of the timed span in which
# no RoIAlign kernel runs, in which two RoIAlign kernels overlap, and the per-kernel dispatch-to-dispatch gaps
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04tl; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o tl -- python bench.py --steps 200 --warmup 10 --no-cpu-baseline --sustain-seconds 0 ${BENCH_ARGS:-} > $O/tl.log 2>&1 < /dev/null
python - <<PY
import csv, collections
rows = [r for r in csv.DictReader(open("$O/tl_kernel_trace.csv"))]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows]
ev.sort()
# the steady state: the last 60 % of the dispatches of the run
ev = ev[int(len(ev) * 0.4):]
t0, t1 = ev[0][0], max(e[1] for e in ev)
span = t1 - t0
def is_ra(n): return "roi_align" in n
pts = []
for s, e, n, q in ev:
    pts.append((s, 1, is_ra(n))); pts.append((e, -1, is_ra(n)))
pts.sort()
ra = other = 0; last = t0
acc = collections.Counter()
for t, d, r in pts:
    key = ("2+ RoIAlign" if ra >= 2 else "1 RoIAlign" if ra == 1 else "no RoIAlign, short kernels only" if other > 0 else "idle")
    acc[key] += t - last; last = t
    if r: ra += d
    else: other += d
n_steps = sum(1 for e in ev if "mask_paste" in e[2])
print("steady-state span %.3f ms, %d steps -> %.4f ms/step" % (span / 1e6, n_steps, span / 1e6 / max(1, n_steps)))
for k, v in acc.most_common(): print("  %-34s %5.1f %%  (%.4f ms/step)" % (k, 100.0 * v / span, v / 1e6 / max(1, n_steps)))
dur = collections.defaultdict(list)
for s, e, n, q in ev: dur[n.split("(")[0][:60]].append((e - s) / 1e3)
print("kernel durations under overlap (us): mean [min .. max]")
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:16]: print("  %-62s %8.1f [%7.1f .. %8.1f] x %d" % (n, sum(v) / len(v), min(v), max(v), len(v)))
PY
