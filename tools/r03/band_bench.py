"""Development harness: the bench's box-head RoIAlign launch through the band-sweep kernel and through the cluster-stationary
kernel on the same descriptors -- bit-exact comparison, launch times (HIP events), band item statistics.
    python tools/r03/band_bench.py [--batch 8] [--iters 30] [--workload cfg3|hard]"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from detectorch_amd import hip  # noqa: E402
from detectorch_amd.pipeline import FpnRegionPath, synthetic_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--tag", default="")
    ap.add_argument("--mask", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    paths = []
    for s in (3000, 4000):
        p = FpnRegionPath(a.batch, dev)
        p.bind(*synthetic_batch(a.batch, dev, seed=s))
        p.step(use_graph=False)
        paths.append(p)
    torch.cuda.synchronize()
    L = hip.lib()
    R = a.batch * paths[0].top_n
    bad = 0
    for p in paths:
        band = p.box_feats.clone()
        ref = torch.empty_like(band)
        hip.check(L.dtc_roi_align_forward_packed(p.feat_lv, 4, p.C, p.feat_code, p.roi_desc.data_ptr(), R, 7, 7, 2, ref.data_ptr(),
                                                 p.out_code, hip.stream_ptr(dev)), "packed")
        torch.cuda.synchronize()
        bad += int((band != ref).any(dim=(1, 2, 3)).sum())
    ctl = paths[0].band_ws[:256].cpu().numpy().view(np.int32)
    print("%s RoIs differing from the cluster kernel: %d of %d ; band items %d gather items %d ; slices %s"
          % (a.tag, bad, 2 * R, ctl[0], ctl[1], [int(x) for x in ctl[24:33]]))

    def t(fn):
        for _ in range(3):
            fn(0); fn(1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(a.iters):
            fn(i & 1)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters

    def band(i):
        paths[i]._roi_align_box()

    def cluster(i):
        p = paths[i]
        L.dtc_roi_align_forward_packed(p.feat_lv, 4, p.C, p.feat_code, p.roi_desc.data_ptr(), R, 7, 7, 2, p.box_feats.data_ptr(),
                                       p.out_code, hip.stream_ptr(dev))
    tb, tc = t(band), t(cluster)
    if a.mask:
        Rm = a.batch * paths[0].max_out
        def mask_ws(i):
            p = paths[i]
            L.dtc_roi_align_forward_banded(p.feat_lv, 4, p.C, p.feat_code, p.m_desc.data_ptr(), Rm, 14, 14, 2, p.mask_feats.data_ptr(),
                                           p.out_code, p.band_ws.data_ptr(), p.band_ws.numel(), hip.stream_ptr(dev))
        def mask_packed(i):
            p = paths[i]
            L.dtc_roi_align_forward_packed(p.feat_lv, 4, p.C, p.feat_code, p.m_desc.data_ptr(), Rm, 14, 14, 2, p.mask_feats.data_ptr(),
                                           p.out_code, hip.stream_ptr(dev))
        ref = paths[0].mask_feats.clone(); mask_ws(0); torch.cuda.synchronize()
        print("%s mask head: workspace entry == packed entry: %s ; %.4f ms vs %.4f ms" % (a.tag, bool(torch.equal(ref, paths[0].mask_feats)), t(mask_ws), t(mask_packed)))
    if hasattr(L, "dtc_debug_band_trace"):          # the development build (tools/r03/build_trace_lib.sh): cycles per phase
        buf = (ctypes.c_ulonglong * 16)()
        L.dtc_debug_band_trace(buf, 1)
        band(0)
        torch.cuda.synchronize()
        L.dtc_debug_band_trace(buf, 1)
        names = ["unit fetch", "first batch (issue+commit)", "set-up", "issue", "pool", "wait B1", "commit", "slab store", "wait B2"]
        tot = float(sum(buf[:9])) or 1.0
        print("phase trace of one launch (thread 0 of every workgroup, shader cycles summed over %d workgroups): total %.1f M" % (256, tot / 1e6))
        for i, nm in enumerate(names):
            print("   %-28s %8.2f M  %5.1f %%" % (nm, buf[i] / 1e6, 100.0 * buf[i] / tot))
    alg = paths[0].box_roialign_bytes()
    print("%s band entry (prep kernels + sweep) %.4f ms = %.2f TB/s (frac %.3f) | cluster kernel %.4f ms (frac %.3f) | env %s"
          % (a.tag, tb, alg / tb / 1e9, alg / tb / 1e9 / 8.0, tc, alg / tc / 1e9 / 8.0,
             " ".join("%s=%s" % kv for kv in sorted(os.environ.items()) if kv[0].startswith("DTC_"))))
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
