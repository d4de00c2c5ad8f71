cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03ae; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_epilogue.py tests/test_hip_detector.py -x -q -m gpu 2>&1 | tail -4 | tee $O/tests.txt
timeout 900 python tools/bench_detector.py --batched --batch 8 --dtype bf16 --channels-last --optimize --miopen-benchmark 2>&1 | tail -1 | cut -c1-360 | tee -a $O/summary.txt
