"""Micro-benchmark of the RoIAlign kernel alone (development aid; bench.py is the contract benchmark)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectorch_amd import hip, synth


def fpn_level_of(rois):
    area = (rois[:, 2] - rois[:, 0] + 1) * (rois[:, 3] - rois[:, 1] + 1)
    lv = np.floor(4 + np.log2(np.sqrt(area) / 224 + 1e-6))
    return (np.clip(lv, 2, 5) - 2).astype(np.int32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--rois", type=int, default=1000)
    ap.add_argument("--pooled", type=int, default=7)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--nhwc", action="store_true")
    ap.add_argument("--half", action="store_true")
    ap.add_argument("--sort", action="store_true", help="visit RoIs sorted by (image, level, row band, x) like dtc_fpn_collect_distribute")
    ap.add_argument("--band-log2", type=int, default=int(os.environ.get("DTC_FPN_BAND_LOG2", "5")))
    ap.add_argument("--max-side", type=float, default=600.0, help="largest RoI side in pixels (48: every window <= ~14x14 on P2)")
    a = ap.parse_args()
    rs = synth.rng(3, 0)
    shapes = synth.fpn_level_shapes()[:4]
    feats = [torch.from_numpy(synth.make_features(rs, (a.batch, a.channels, h, w))).cuda() for (h, w) in shapes]
    if a.half:
        feats = [f.half() for f in feats]
    if a.nhwc:
        feats = [f.contiguous(memory_format=torch.channels_last) for f in feats]
    rois = np.concatenate([np.hstack([np.full((a.rois, 1), b, np.float32), synth.make_rois(rs, a.rois, max_side=a.max_side)])
                           for b in range(a.batch)])
    lvn = fpn_level_of(rois[:, 1:])
    order = None
    if a.sort:
        yc, xc = (rois[:, 2] + rois[:, 4]) * 0.5, (rois[:, 1] + rois[:, 3]) * 0.5
        band = (yc / (4.0 * 2.0 ** lvn * (1 << a.band_log2))).astype(np.int32)       # 2^band_log2 feature rows of the level
        order = torch.from_numpy(np.lexsort((xc, band, lvn, rois[:, 0])).astype(np.int32)).cuda()
    lv = torch.from_numpy(lvn).cuda()
    print("level histogram:", np.bincount(lv.cpu().numpy(), minlength=4))
    rois = torch.from_numpy(rois).cuda()
    odt = torch.float16 if a.half else torch.float32
    out = torch.empty((rois.shape[0], a.channels, a.pooled, a.pooled), dtype=odt, device="cuda")
    for _ in range(3):
        hip.roi_align_forward(feats, synth.FPN_ROI_SCALES, rois, a.pooled, a.pooled, 2, roi_levels=lv, out=out, roi_order=order)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        hip.roi_align_forward(feats, synth.FPN_ROI_SCALES, rois, a.pooled, a.pooled, 2, roi_levels=lv, out=out, roi_order=order)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    fbytes = sum(f.numel() * f.element_size() for f in feats)
    obytes = out.numel() * out.element_size()
    alg = fbytes + obytes + rois.numel() * 4
    print("batch %d rois/img %d pooled %d C %d %s %s: %.3f ms/launch = %.1f us/img ; algorithmic %.1f MB -> %.2f TB/s"
          % (a.batch, a.rois, a.pooled, a.channels, "NHWC" if a.nhwc else "NCHW", "fp16" if a.half else "fp32", ms,
             ms * 1e3 / a.batch, alg / 1e6, alg / ms / 1e9))


if __name__ == "__main__":
    main()
