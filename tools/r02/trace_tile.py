"""Development aid: per-phase cycle accounting of the cluster RoIAlign kernel on the bench box-head launch.
   bash tools/r02/build_trace_lib.sh   (here)   then on the GPU box:   python tools/r02/trace_tile.py [--mask]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DETECTORCH_HIP_LIB"] = os.path.join(ROOT, "detectorch_amd", "lib", "trace", "libdetectorch_hip.so")
sys.path.insert(0, ROOT)
import numpy as np
import torch
from detectorch_amd import hip
from detectorch_amd.pipeline import FpnRegionPath, synthetic_batch

mask = "--mask" in sys.argv
dev = torch.device("cuda", 0)
path = FpnRegionPath(8, dev)
path.bind(*synthetic_batch(8, dev, seed=3000))
path.step(use_graph=False)
torch.cuda.synchronize()
fn = path._roi_align_mask if mask else path._roi_align_box
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record(); torch.cuda.synchronize()
print("launch %.4f ms" % e0.elapsed_time(e1))
L = hip.lib()
buf = np.zeros((16384, 16), dtype=np.uint64)
L.dtc_debug_tile_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert L.dtc_debug_tile_trace(buf.ctypes.data, buf.nbytes) == 0
n = int((buf[:, 12] > 0).sum())
t = buf[:n].astype(np.float64)
names = ["A windows", "B cluster", "C item setup", "uoff setup", "commit(+load wait)", "store_slab", "barrier1", "issue", "pool", "(mark8)", "barrier2"]
names = ["A windows", "B cluster", "C item setup", "unit setup", "commit(+load wait)", "store_slab", "barrier1", "issue", "pool", "barrier2", "final store"]
tot = t[:, :11].sum(1)
print("workgroups %d ; cycles per workgroup: mean %.0f  p50 %.0f  p90 %.0f" % (n, tot.mean(), np.median(tot), np.percentile(tot, 90)))
for i, nm in enumerate(names):
    print("  %-20s %8.0f cyc  %5.1f %%" % (nm, t[:, i].mean(), 100 * t[:, i].sum() / tot.sum()))
wall = (t[:, 12] - t[:, 11]) / 100.0   # us (100 MHz)
span = (t[:, 12].max() - t[:, 11].min()) / 100.0
print("wall per workgroup: mean %.2f us ; kernel span %.1f us ; mean concurrency %.1f workgroups (%.2f per CU)" % (wall.mean(), span, wall.sum() / span, wall.sum() / span / 256))
print("clusters per workgroup: mean %.2f" % t[:, 13].mean())
# start-time profile: how many workgroups start in each tenth of the span
st = (t[:, 11] - t[:, 11].min()) / 100.0
print("starts per decile:", np.histogram(st, bins=10, range=(0, span))[0].tolist())
# per-XCD balance (block b runs on XCD b % 8): when does each XCD finish, how long are its workgroups
end = (t[:, 12] - t[:, 11].min()) / 100.0
xcd = np.arange(n) % 8
print("per XCD: finish us / mean workgroup us / sum workgroup-us")
for x in range(8):
    m = xcd == x
    print("  xcd %d: %6.1f  %6.2f  %8.0f" % (x, end[m].max(), wall[m].mean(), wall[m].sum()))
# the slowest workgroups: where in the visiting order are they
o = np.argsort(-wall)[:12]
print("slowest:", [(int(t[i, 15]), round(float(wall[i]), 1), int(t[i, 13])) for i in o])
# duration by position in the visiting order (deciles of work item index)
wi = t[:, 15]
for d in range(10):
    m = (wi >= d * n / 10) & (wi < (d + 1) * n / 10)
    print("  items %4.0f%%: mean %.1f us  p90 %.1f us" % (d * 10, wall[m].mean(), np.percentile(wall[m], 90)))
# share of workgroup time by FPN level (work item wi -> cluster group wi // nct -> its 5 RoIs in visiting order)
if not mask:
    desc = path.roi_desc.reshape(-1, 8).cpu().numpy()
    nct = 4
    K = 5
    grp = (t[:, 15] // nct).astype(np.int64)
    lv = np.array([int(np.median(desc[g * K:(g + 1) * K, 5])) for g in grp])
    print("share of workgroup time by level (level index 0 = P2):")
    for l in sorted(set(lv.tolist())):
        m = lv == l
        print("   level %2d: %5d workgroups (%.1f %%)  time share %.1f %%  mean %.1f us" % (l, m.sum(), 100 * m.mean(), 100 * wall[m].sum() / wall.sum(), wall[m].mean()))
