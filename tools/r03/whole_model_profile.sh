cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03ac; mkdir -p $O
timeout 900 python tools/r03/prof_batched.py opt bench > $O/prof_opt_bench.txt 2>&1
for args in "--dtype bf16 --channels-last --miopen-benchmark" "--dtype fp16 --channels-last --optimize --miopen-benchmark" "--dtype fp32 --optimize --miopen-benchmark" "--dtype fp32 --channels-last --optimize --miopen-benchmark"; do
  timeout 900 python tools/bench_detector.py --batched --batch 8 $args 2>&1 | tail -1 | cut -c1-360 | tee -a $O/summary.txt
done
