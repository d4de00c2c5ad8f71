cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03q; mkdir -p $O
for a in "--batch 8 --rois 1000" "--batch 4 --rois 2000" "--batch 2 --rois 4000" "--batch 1 --rois 8000"; do
  for ms in 90 600; do
    echo -n "$a max-side $ms: " | tee -a $O/summary.txt; timeout 200 python tools/bench_roialign.py --sort --max-side $ms $a 2>/dev/null | tail -1 | tee -a $O/summary.txt
  done
done
