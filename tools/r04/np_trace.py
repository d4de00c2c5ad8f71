"""Phase trace of the pipelined channels_last RoIAlign kernel (needs the -DDTC_NP_TRACE build: tools/r04/build_trace_lib.sh,
DETECTORCH_HIP_LIB=detectorch_amd/lib/trace/libdetectorch_hip.so).  Runs the bench's box-head launch once and prints, summed over
the workgroups, the share of thread 0's cycles per phase."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detectorch_amd import hip
from detectorch_amd.pipeline import FpnRegionPath, synthetic_batch

dev = torch.device("cuda", 0)
fp16 = "--fp16" in sys.argv
fdt = torch.float16 if fp16 else torch.float32
path = FpnRegionPath(8, dev, feat_dtype=fdt)
path.bind(*synthetic_batch(8, dev, seed=3000, feat_dtype=fdt, channels_last=True))
path.step(use_graph=False)
torch.cuda.synchronize()
for _ in range(3):
    path._roi_align_box()
torch.cuda.synchronize()
L = hip.lib()
buf = np.zeros((4096, 2, 16), np.uint64)
L.dtc_debug_np_trace.argtypes = [C.c_void_p, C.c_size_t]
assert L.dtc_debug_np_trace(buf.ctypes.data, buf.nbytes) == 0
used = buf[buf[:, 0, 8] > 0]
pl, po = used[:, 0], used[:, 1]
units = pl[:, 8].sum()
print("workgroups traced: %d ; units %d, items %d, staged pixels %d" % (len(used), units, pl[:, 9].sum() + len(used), pl[:, 10].sum()))
for who, arr, names in (("planner (wave 0)", pl, ["loop top", "-", "T barrier (waits for the poolers)", "X barrier", "plan (+ item set-up, tables)"]),
                        ("pooler (wave 1)", po, ["loop top", "wait vmcnt(0) + T barrier", "flush + descriptor + DMA issue", "-", "pool (+ stores of all rounds but the last)"])):
    tot = arr[:, :8].sum()
    print("%s: %.0f cycles per workgroup, %.0f per unit" % (who, tot / max(1, len(used)), tot / max(1, units)))
    for i, n in enumerate(names):
        print("  %-28s %5.1f %%   %8.0f cycles per unit" % (n, 100.0 * arr[:, i].sum() / tot, arr[:, i].sum() / max(1, units)))
