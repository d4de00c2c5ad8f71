cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03z; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_epilogue.py tests/test_hip_detector.py -x -q -m gpu 2>&1 | tail -5 | tee $O/tests.txt
for args in "--dtype bf16 --channels-last --optimize" "--dtype fp16 --channels-last --optimize" "--dtype bf16 --optimize" "--dtype fp32 --optimize" "--dtype fp32 --channels-last --optimize"; do
  timeout 600 python tools/bench_detector.py --batched --batch 8 $args 2>&1 | tail -1 | cut -c1-360 | tee -a $O/summary.txt
done
timeout 600 python tools/r03/prof_batched.py opt > $O/prof_opt.txt 2>&1
