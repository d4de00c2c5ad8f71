#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_hip_roi_align.py -x -q -k "map" 2>&1 | tail -5
python - <<'PY'
import sys, torch, numpy as np
sys.path.insert(0, ".")
from detectorch_amd import hip
from detectorch_amd.pipeline import C4RegionPath, synthetic_c4_batch
dev = torch.device("cuda", 0)
p = C4RegionPath(8, dev, pooled=7); p.bind(*synthetic_c4_batch(8, dev, seed=2000))
p.step(use_graph=False); torch.cuda.synchronize()
exact = p.box_feats.clone()
def t(n=20):
    for _ in range(3): p._roi_align_box()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): p._roi_align_box()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("exact mode: %.4f ms per launch" % t())
hip.roi_align_set_exact(False)
print("fast mode : %.4f ms per launch" % t())
d = (p.box_feats - exact).abs().max().item()
print("max |fast - exact| = %.3g (max |x| %.3g)" % (d, exact.abs().max().item()))
hip.roi_align_set_exact(True)
PY
