"""Phase trace of the cluster kernel (roi_align_fwd_tile built with -DDTC_TILE_TRACE: tools/r06/tile_trace.sh): wave 0 of every workgroup
accumulates its cycle counter per phase; this prints the mean per workgroup and the share of each phase for ONE launch.

    DETECTORCH_HIP_LIB=$PWD/detectorch_amd/lib/trace/libdetectorch_hip.so python tools/r06/tile_trace.py [--mask] [--harder] [--fp16 --top-n 2000]
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detectorch_amd import hip, synth  # noqa: E402
from detectorch_amd.pipeline import FpnRegionPath, synthetic_batch  # noqa: E402

NAMES = ["A+B windows, clusters (wave 0) + barrier", "(unused)", "item set-up (per cluster)", "staging-unit set-up", "commit", "slab store", "barrier 1",
         "issue", "pool", "barrier 2", "last slab store"]
ap = argparse.ArgumentParser()
ap.add_argument("--mask", action="store_true")
ap.add_argument("--harder", action="store_true")
ap.add_argument("--fp16", action="store_true")
ap.add_argument("--top-n", type=int, default=1000)
a = ap.parse_args()
dev = torch.device("cuda", 0)
fdt = torch.float16 if a.fp16 else torch.float32
path = FpnRegionPath(8, dev, feat_dtype=fdt, collect_top_n=a.top_n)
path.bind(*synthetic_batch(8, dev, seed=3000, feat_dtype=fdt, top_n=a.top_n))
path.step(use_graph=False)
torch.cuda.synchronize()
fn = path._roi_align_mask if a.mask else path._roi_align_box
if a.harder:
    rois, lvn, order = synth.harder_roi_set(8, a.top_n)
    rois_t, lv, od = (torch.from_numpy(x).to(dev) for x in (rois, lvn, order))
    P = path.mask_p if a.mask else path.box_p
    out = torch.empty((rois.shape[0], path.C, P, P), dtype=fdt, device=dev)
    fn = lambda: hip.roi_align_forward(path.feats, synth.FPN_ROI_SCALES, rois_t, P, P, 2, roi_levels=lv, out=out, roi_order=od)
for _ in range(3):
    fn()
torch.cuda.synchronize()
fn()
torch.cuda.synchronize()
n = 16384
buf = np.zeros((n, 16), np.uint64)
lib = hip.lib()
rc = lib.dtc_debug_tile_trace(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
used = buf[:, 12] > 0                       # end wall clock written
st = np.sort(buf[used, 11])                 # ... by the LAST launch (earlier, larger launches leave their rows behind): the rows behind the
gaps = np.flatnonzero(np.diff(st.astype(np.int64)) > 10000)          # last gap of more than 100 us between workgroup starts
if gaps.size:
    used &= buf[:, 11] > st[gaps[-1]]
t = buf[used]
ph = t[:, :11].astype(np.float64)
tot = ph.sum(1)
wall = (t[:, 12] - t[:, 11]).astype(np.float64)          # s_memrealtime ticks (100 MHz)
print("%d workgroups traced; mean %.0f shader cycles per workgroup (wave 0), wall %.2f us per workgroup, clusters per workgroup %.2f" %
      (used.sum(), tot.mean(), wall.mean() / 100.0, t[:, 13].astype(np.float64).mean()))
for i, nm in enumerate(NAMES):
    if ph[:, i].sum() > 0:
        print("  %-44s %9.0f cycles  %5.1f %%" % (nm, ph[:, i].mean(), 100.0 * ph[:, i].sum() / tot.sum()))
span = (t[:, 12].max() - t[:, 11].min()) / 100.0
print("launch span (first start .. last end of the traced workgroups): %.1f us" % span)
