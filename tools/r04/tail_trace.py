"""Phase trace of the short kernels of one step (needs tools/r04/tail_trace.sh's build, DETECTORCH_HIP_LIB=<trace lib>): thread 0 of
the first 64 workgroups of each instrumented kernel stamps the 100 MHz clock at its phase marks; prints, per kernel, the mean and
max time between consecutive marks and from the first mark to the last."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detectorch_amd import hip
from detectorch_amd.pipeline import FpnRegionPath, synthetic_batch
dev = torch.device("cuda", 0)
path = FpnRegionPath(8, dev, max_out=104); path.bind(*synthetic_batch(8, dev, seed=3000, max_out=104))
for _ in range(3): path.step(use_graph=False)
torch.cuda.synchronize()
L = hip.lib()
names = {"proposals": {0: "rpn_hist<0>", 1: "rpn_hist<1>", 2: "rpn_compact", 3: "rpn_sort"}, "detections": {0: "det_candidates", 1: "det_finalize"},
         "nms": {0: "nms_reduce_lds"}, "fpn": {0: "fpn_collect_fast (box)", 1: "fpn_collect_fast (mask)"}, "mask_paste": {0: "mask_paste", 1: "mask_paste (helpers)"}}
for f, ks in names.items():
    buf = np.zeros((4, 64, 24), np.uint64)
    fn = getattr(L, "dtc_debug_phase_trace_" + f); fn.argtypes = [C.c_void_p, C.c_size_t]
    assert fn(buf.ctypes.data, buf.nbytes) == 0
    for k, nm in ks.items():
        t = buf[k].astype(np.int64)
        used = t[t[:, 0] > 0]
        if not len(used): continue
        marks = [m for m in range(24) if (used[:, m] > 0).all()]
        segs = ["%d->%d %.1f (max %.1f)" % (a, b, (used[:, b] - used[:, a]).mean() / 100.0, (used[:, b] - used[:, a]).max() / 100.0) for a, b in zip(marks[:-1], marks[1:])]
        print("%-26s first->last %.1f us mean, %.1f max | %s" % (nm, (used[:, marks[-1]] - used[:, marks[0]]).mean() / 100.0, (used[:, marks[-1]] - used[:, marks[0]]).max() / 100.0, " ; ".join(segs)))
