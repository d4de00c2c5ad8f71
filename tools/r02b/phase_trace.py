"""Development aid: wall-clock phase stamps (100 MHz) of the latency-bound kernels of one eager step at the bench configuration.
   bash tools/r02b/build_phase_trace_lib.sh (here), then on the GPU box: python tools/r02b/phase_trace.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DETECTORCH_HIP_LIB"] = os.path.join(ROOT, "detectorch_amd", "lib", "ptrace", "libdetectorch_hip.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
from detectorch_amd import hip
from detectorch_amd.pipeline import FpnRegionPath, synthetic_batch
dev = torch.device("cuda", 0)
path = FpnRegionPath(8, dev)
path.bind(*synthetic_batch(8, dev, seed=3000))
for _ in range(3):
    path.step(use_graph=False)
torch.cuda.synchronize()
L = hip.lib()
K, B, M = 4, 64, 24
tables = {"proposals": ["rpn_hist<0>", "rpn_hist<1>", "rpn_compact", "rpn_sort_decode"],
          "detections": ["det_candidates", "det_finalize", None, None], "fpn": ["fpn_fast(box)", "fpn_fast(mask)", None, None],
          "mask_paste": ["paste(main)", "paste(helper)", None, None],
          "nms": ["nms_reduce(lds)", "nms_reduce(1 wave)", None, None]}
for name, kernels in tables.items():
    fn = getattr(L, "dtc_debug_phase_trace_" + name)
    fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    buf = np.zeros((K, B, M), dtype=np.uint64)
    assert fn(buf.ctypes.data, buf.nbytes) == 0
    for k, kn in enumerate(kernels):
        if kn is None: continue
        t = buf[k].astype(np.float64)
        used = t[:, 0] > 0
        if not used.any(): continue
        t = t[used]
        nm = int((t[0] > 0).sum())
        # marks may be missing for workgroups that returned early: use blocks that reached the last mark
        full = t[:, nm - 1] > 0
        tt = t[full][:, :nm]
        d = np.diff(tt, axis=1) / 100.0
        print("%-16s blocks %2d  total %6.2f us (max %6.2f) | phases us: %s" % (kn, full.sum(), (tt[:, -1] - tt[:, 0]).mean() / 100.0,
              (tt[:, -1] - tt[:, 0]).max() / 100.0, "  ".join("%.2f" % x for x in d.mean(0))))
