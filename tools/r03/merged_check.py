"""merged-tap pooling of the cluster kernel against its exact mode: max |difference| on FPN-distributed RoIs (development aid)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detectorch_amd import hip, synth
rs = synth.rng(3, 0)
shapes = synth.fpn_level_shapes()[:4]
for C, ph, dt in [(256, 7, torch.float32), (256, 14, torch.float32), (64, 7, torch.float16)]:
    feats = [torch.from_numpy(synth.make_features(rs, (2, C, h, w))).cuda().to(dt) for (h, w) in shapes]
    for max_side in (48.0, 120.0, 600.0):
        rois = np.concatenate([np.hstack([np.full((500, 1), b, np.float32), synth.make_rois(rs, 500, max_side=max_side)]) for b in range(2)])
        area = (rois[:, 3] - rois[:, 1] + 1) * (rois[:, 4] - rois[:, 2] + 1)
        lv = (np.clip(np.floor(4 + np.log2(np.sqrt(area) / 224 + 1e-6)), 2, 5) - 2).astype(np.int32)
        yc, xc = (rois[:, 2] + rois[:, 4]) * 0.5, (rois[:, 1] + rois[:, 3]) * 0.5
        band = (yc / (4 * 2 ** lv.astype(np.float32) * 16)).astype(np.int32)
        order = np.lexsort((xc, band, lv, rois[:, 0]))
        r = torch.from_numpy(rois[order]).cuda(); l = torch.from_numpy(lv[order]).cuda()
        with hip.roi_align_exact(True):
            ex = hip.roi_align_forward(feats, synth.FPN_ROI_SCALES, r, ph, ph, 2, roi_levels=l).float()
        with hip.roi_align_exact(False):
            mg = hip.roi_align_forward(feats, synth.FPN_ROI_SCALES, r, ph, ph, 2, roi_levels=l).float()
        d = (ex - mg).abs()
        print("C %d bins %dx%d %s max_side %.0f: max|exact| %.3f  max abs diff %.3e  mean abs diff %.3e  differing %.1f %%" %
              (C, ph, ph, str(dt).split('.')[-1], max_side, ex.abs().max().item(), d.max().item(), d.mean().item(), 100.0 * (d > 0).float().mean().item()))
