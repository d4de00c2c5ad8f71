"""Development aid: time the BENCH workload's box-head (and mask-head) RoIAlign launch alone -- the pipeline runs once to
produce the real visiting order / descriptors, then only the RoIAlign launch is repeated.  bench.py is the contract."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectorch_amd.pipeline import FpnRegionPath, synthetic_batch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--fp16", action="store_true")
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--mask", action="store_true", help="time the 14x14 mask-head launch instead")
    ap.add_argument("--channels-last", action="store_true", help="NHWC feature maps (same logical shape)")
    ap.add_argument("--top-n", type=int, default=1000, help="RoIs per image after collect (cfg5: 2000)")
    ap.add_argument("--harder", action="store_true", help="the harder RoI population (synth.harder_roi_set) instead of the path's own proposals")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    fdt = torch.float16 if a.fp16 else torch.bfloat16 if a.bf16 else torch.float32
    path = FpnRegionPath(a.batch, dev, feat_dtype=fdt, collect_top_n=a.top_n)
    path.bind(*synthetic_batch(a.batch, dev, seed=3000, feat_dtype=fdt, top_n=a.top_n, channels_last=a.channels_last))
    path.step(use_graph=False)
    torch.cuda.synchronize()
    fn = path._roi_align_mask if a.mask else path._roi_align_box
    if a.harder:
        from detectorch_amd import hip, synth
        rois, lvn, order = synth.harder_roi_set(a.batch, a.top_n)
        rois_t, lv, od = (torch.from_numpy(x).to(dev) for x in (rois, lvn, order))
        P = path.mask_p if a.mask else path.box_p
        out = torch.empty((rois.shape[0], path.C, P, P), dtype=fdt, device=dev)
        fn = lambda: hip.roi_align_forward(path.feats, synth.FPN_ROI_SCALES, rois_t, P, P, 2, roi_levels=lv, out=out, roi_order=od)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    alg = path.box_roialign_bytes()
    print("%s-head RoIAlign, batch %d: %.4f ms/launch%s" % ("mask" if a.mask else "box", a.batch, ms,
          "" if a.mask else " ; algorithmic %.1f MB -> %.2f TB/s (frac %.3f of 8 TB/s)" % (alg / 1e6, alg / ms / 1e9, alg / ms / 1e9 / 8)))


if __name__ == "__main__":
    main()
