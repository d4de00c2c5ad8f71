cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03w; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_epilogue.py -x -q -m gpu 2>&1 | tail -5 | tee $O/tests.txt
timeout 300 python tools/bench_epilogue.py 2>&1 | grep -v amdgpu.ids | tee $O/epilogue.txt
for args in "--dtype bf16 --channels-last --optimize" "--dtype bf16 --optimize"; do
  timeout 600 python tools/bench_detector.py --batched --batch 8 $args 2>&1 | tail -1 | cut -c1-330 | tee -a $O/summary.txt
done
