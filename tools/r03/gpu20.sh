cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03y; mkdir -p $O
timeout 600 python tools/r03/prof_batched.py opt > $O/prof_opt.txt 2>&1
