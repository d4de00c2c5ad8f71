"""The C4 (cfg2) RoIAlign launch alone -- roi_align_fwd_map on 8 x 1000 RoIs x 1024 channels, adaptive sampling -- in exact or fast mode
(counter runs: tools/r06/counters.sh).   python tools/r06/c4_launch.py --mode exact|fast [--iters N]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detectorch_amd import hip  # noqa: E402
from detectorch_amd.pipeline import C4RegionPath, synthetic_c4_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mode", choices=["exact", "fast"], default="exact")
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda", 0)
hip.roi_align_set_exact(a.mode == "exact")
path = C4RegionPath(8, dev)
path.bind(*synthetic_c4_batch(8, dev, seed=2000))
path.step(use_graph=False)
torch.cuda.synchronize()
for _ in range(3):
    path._roi_align_box()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    path._roi_align_box()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
print("%s: %.4f ms per launch, frac %.4f of 8 TB/s" % (a.mode, ms, path.box_roialign_bytes() / ms / 1e6 / 8000))
