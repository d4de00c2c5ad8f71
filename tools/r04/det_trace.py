import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detectorch_amd import hip
from detectorch_amd.pipeline import FpnRegionPath, synthetic_batch
dev = torch.device("cuda", 0)
path = FpnRegionPath(8, dev); path.bind(*synthetic_batch(8, dev, seed=3000))
for _ in range(3): path.step(use_graph=False)
torch.cuda.synchronize()
L = hip.lib()
buf = np.zeros((4, 64, 24), np.uint64)
L.dtc_debug_phase_trace_detections.argtypes = [C.c_void_p, C.c_size_t]
assert L.dtc_debug_phase_trace_detections(buf.ctypes.data, buf.nbytes) == 0
t = buf[0].astype(np.int64)
print("n per traced segment:", t[:, 16][:32])
marks = [0, 1, 2, 3, 4, 5, 6, 7, 15]
for i in range(len(marks) - 1):
    a, b = marks[i], marks[i + 1]
    ok = (t[:, a] > 0) & (t[:, b] > 0)
    print("mark %2d -> %2d : %6.2f us mean, %6.2f max" % (a, b, (t[ok, b] - t[ok, a]).mean() / 100.0, (t[ok, b] - t[ok, a]).max() / 100.0))
print("total 0 -> 15: %.2f us mean" % ((t[:, 15] - t[:, 0]).mean() / 100.0), "; spread of starts %.2f us" % ((t[:, 0].max() - t[:, 0].min()) / 100.0))
t1 = buf[1].astype(np.int64)
print("det_finalize marks (us from mark 0), mean over images:", [round(float((t1[:8, i] - t1[:8, 0]).mean()) / 100.0, 2) for i in range(6)])
