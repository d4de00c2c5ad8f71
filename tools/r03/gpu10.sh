cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03n; mkdir -p $O
for kb in 32 36 40 46 52 60; do
  echo -n "KB=$kb fp32: " | tee -a $O/summary.txt; DTC_RA_NHWC_LDS_KB=$kb timeout 200 python tools/bench_roialign.py --nhwc --sort --max-side 90 2>/dev/null | tail -1 | tee -a $O/summary.txt
done
for kb in 40 52; do
  echo -n "KB=$kb fp16: " | tee -a $O/summary.txt; DTC_RA_NHWC_LDS_KB=$kb timeout 200 python tools/bench_roialign.py --nhwc --half --sort --max-side 90 2>/dev/null | tail -1 | tee -a $O/summary.txt
done
