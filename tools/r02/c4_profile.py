"""Development aid: the C4 (cfg2) RoIAlign launch by RoI size class -- window pixels, adaptive grid, time per class."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from detectorch_amd import hip
from detectorch_amd.pipeline import C4RegionPath, synthetic_c4_batch

dev = torch.device("cuda", 0)
B = 8
path = C4RegionPath(B, dev)
inp = synthetic_c4_batch(B, dev, seed=2000)
path.bind(*inp)
path.step(use_graph=False)
torch.cuda.synchronize()
feat = inp[2]
rois5 = path.rois5.reshape(-1, 5).clone()
n = path.n_rois.tolist()
valid = torch.cat([torch.arange(b * path.top_n, b * path.top_n + n[b]) for b in range(B)]).to(dev)
r = rois5[valid]
w = ((r[:, 3] - r[:, 1]) / 16).clamp(min=1.0).cpu().numpy()
h = ((r[:, 4] - r[:, 2]) / 16).clamp(min=1.0).cpu().numpy()
gw, gh = np.ceil(w / 7), np.ceil(h / 7)
win = (np.floor(w) + 2) * (np.floor(h) + 2)
print("valid rois %d ; grid product gh*gw: mean %.2f ; window px: mean %.0f p50 %.0f p90 %.0f max %.0f" % (len(w), (gw * gh).mean(), win.mean(), np.median(win), np.percentile(win, 90), win.max()))
for lo, hi in [(0, 64), (64, 256), (256, 512), (512, 1024), (1024, 2048), (2048, 1e9)]:
    m = (win >= lo) & (win < hi)
    print("  window [%5.0f, %5.0f): %5d rois (%.1f %%)  mean samples/bin %.1f" % (lo, hi, m.sum(), 100 * m.mean(), (gw * gh)[m].mean() if m.any() else 0))


def time_subset(mask, tag, order=None):
    sel = np.nonzero(mask)[0]
    if order is not None:
        sel = sel[order]
    idx = valid[torch.from_numpy(sel).to(dev)]
    rr = rois5[idx].contiguous()
    if len(rr) == 0:
        return
    for _ in range(2):
        out = hip.roi_align_forward(feat, 1.0 / 16, rr, 7, 7, 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        out = hip.roi_align_forward(feat, 1.0 / 16, rr, 7, 7, 0, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("  %-22s %5d rois: %.3f ms  (%.3f us / roi)" % (tag, len(rr), ms, 1e3 * ms / len(rr)))


print("time by class (unsorted order within a class):")
for lo, hi in [(0, 64), (64, 256), (256, 512), (512, 1024), (1024, 2048), (2048, 1e9)]:
    time_subset((win >= lo) & (win < hi), "window [%d, %d)" % (lo, hi))
time_subset(np.ones(len(win), bool), "all")
time_subset(np.ones(len(win), bool), "all, largest first", order=np.argsort(-win, kind="stable"))
time_subset(np.ones(len(win), bool), "all, smallest first", order=np.argsort(win, kind="stable"))


def time_packed(mask, tag):
    """the same subsets through the packed-descriptor entry (image-major order) -> the map-stationary kernel"""
    sel = np.nonzero(mask)[0]
    idx = valid[torch.from_numpy(sel).to(dev)]
    rr = rois5[idx]
    o = torch.argsort(rr[:, 0], stable=True)
    rr = rr[o]
    n_ = len(rr)
    if n_ == 0:
        return
    desc = torch.zeros((n_, 8), device=dev)
    desc[:, :5] = rr
    desc[:, 6] = torch.arange(n_, device=dev, dtype=torch.float32)
    out = torch.empty((n_, feat.shape[1], 7, 7), device=dev)
    lvs, ch, dt = hip.make_levels([feat], [1.0 / 16])
    call = lambda: hip.check(hip.lib().dtc_roi_align_forward_packed(lvs, 1, ch, 0, desc.data_ptr(), n_, 7, 7, 0, out.data_ptr(), 0, hip.stream_ptr()), "packed")
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("  packed %-22s %5d rois: %.3f ms  (%.3f us / roi)" % (tag, n_, ms, 1e3 * ms / n_))


print("map-stationary kernel by class:")
for lo, hi in [(0, 64), (64, 256), (256, 512), (512, 1024), (1024, 2048), (2048, 1e9)]:
    time_packed((win >= lo) & (win < hi), "window [%d, %d)" % (lo, hi))
time_packed(np.ones(len(win), bool), "all")
