cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03ab; mkdir -p $O
for args in "--dtype bf16 --channels-last --optimize --miopen-benchmark" "--dtype bf16 --optimize --miopen-benchmark"; do
  timeout 900 python tools/bench_detector.py --batched --batch 8 $args 2>&1 | tail -2 | cut -c1-360 | tee -a $O/summary.txt
done
