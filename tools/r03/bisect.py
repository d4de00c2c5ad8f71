import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from detectorch_amd import hip
from detectorch_amd.pipeline import FpnRegionPath, synthetic_batch
dev = torch.device("cuda", 0)
p = FpnRegionPath(2, dev, channels=16)
p.bind(*synthetic_batch(2, dev, seed=3000, channels=16))
p.step(use_graph=False)
torch.cuda.synchronize()
ws = p.band_ws.cpu().numpy()
ctl = ws[:256].view(np.int32)
print("stop", os.environ.get("DTC_RA_BAND_STOP"), "ok; n_items", ctl[0], "n_gather", ctl[1], "slice_first", ctl[8:17], "slice_count", ctl[24:33], "ctr", ctl[40:49])
if os.environ.get("DTC_RA_BAND_STOP") == "2":
    n = 2000 + 1
    al = lambda v: (v + 255) & ~255
    off = al(192) + 0
    items = ws[256:256 + 32 * 60].view(np.int32).reshape(-1, 8)
    print("items (first, count, b, lvl, rbase, rows, kind, nbatch):")
    print(items[:ctl[0] + ctl[1]][:60])
    nit = 2 * n
    boff = 256 + al(nit * 32)
    bat = ws[boff:boff + 16 * 2000].view(np.int32).reshape(-1, 4)
    it0 = items[0]
    print("batches of item 0 (i0, n, xa, xb):"); print(bat[it0[0]:it0[0] + it0[7]])
