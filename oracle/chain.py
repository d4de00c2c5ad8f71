"""The whole FPN hot path for ONE image on the CPU, composed from the oracle restatements (oracle.py / oracle.c).

TEST INFRASTRUCTURE ONLY: the checker for detectorch_amd.pipeline.FpnRegionPath (tests/, smoke()) and the thing timed by
bench.py's cpu_baseline leg.  Follows lib/model/detector.py:240-270 + eval_mask_FPN.ipynb:231-262 of the reference.
"""
import time

import numpy as np

import oracle as orc

FPN_STRIDES = (4.0, 8.0, 16.0, 32.0, 64.0)
ROI_SCALES = (0.25, 0.125, 0.0625, 0.03125)


def fpn_hot_path(rpn_cls, rpn_bbox, feats, cls_score, bbox_pred, masks, sf, im_size, pad_h, pad_w, pre=1000, post=1000,
                 top_n=1000, max_det=100, M=28, box_p=7, mask_p=14, sr=2, timings=None, roi_align=None):
    """rpn_cls/rpn_bbox: 5 arrays [A,H,W]/[4A,H,W]; feats: 4 arrays [1,C,H,W]; cls_score [top_n,81]; bbox_pred [top_n,324];
    masks [>=D,81,M,M].  Returns a dict of every intermediate the GPU path produces."""
    t = time.perf_counter
    T = {} if timings is None else timings
    ra = roi_align or orc.roi_align_forward       # bench.py passes the reference-compiled loop (oracle/_ref) when it is there
    t0 = t()
    props, scores = [], []
    for l in range(5):
        anchors = orc.generate_anchors(FPN_STRIDES[l], (32.0 * 2 ** l,), (0.5, 1, 2))
        b, s = orc.generate_proposals(rpn_cls[l], rpn_bbox[l], anchors, FPN_STRIDES[l], pad_h, pad_w, pre, post, 0.7)
        props.append(b); scores.append(s)
    T["generate_proposals"] = T.get("generate_proposals", 0) + t() - t0; t0 = t()
    rois, rsc, _ = orc.collect(np.concatenate(props), np.concatenate(scores), top_n)
    per_level, restore, lv = orc.distribute(rois, 2, 5)
    T["collect_distribute"] = T.get("collect_distribute", 0) + t() - t0; t0 = t()
    n = rois.shape[0]
    C = feats[0].shape[1]
    rois5 = np.hstack([np.zeros((n, 1), np.float32), rois])
    box_feats = np.zeros((n, C, box_p, box_p), np.float32)
    for l in range(4):
        m = lv == l + 2
        if m.any():
            box_feats[m] = ra(feats[l], rois5[m], box_p, box_p, ROI_SCALES[l], sr)
    T["roi_align_box"] = T.get("roi_align_box", 0) + t() - t0; t0 = t()
    dets, det_roi = orc.postprocess_detections(rois, sf, im_size, cls_score[:n], bbox_pred[:n], max_det=max_det)
    T["postprocess"] = T.get("postprocess", 0) + t() - t0; t0 = t()
    D = dets.shape[0]
    scaled = (dets[:, :4] * np.float32(sf)).astype(np.float32)
    mlv = orc.map_rois_to_fpn_levels(scaled, 2, 5)
    m5 = np.hstack([np.zeros((D, 1), np.float32), scaled])
    mask_feats = np.zeros((D, C, mask_p, mask_p), np.float32)
    for l in range(4):
        m = mlv == l + 2
        if m.any():
            mask_feats[m] = ra(feats[l], m5[m], mask_p, mask_p, ROI_SCALES[l], sr)
    T["roi_align_mask"] = T.get("roi_align_mask", 0) + t() - t0; t0 = t()
    crops, boxes = [], []
    for d in range(D):
        bx, crop = orc.mask_resize_binarize(masks[d, int(dets[d, 5])], dets[d, :4])
        boxes.append(bx); crops.append(crop)
    T["mask_paste"] = T.get("mask_paste", 0) + t() - t0
    return dict(props=props, scores=scores, rois=rois, roi_scores=rsc, roi_levels=lv - 2, restore=restore,
                box_feats=box_feats, dets=dets, det_roi=det_roi, det_scaled=scaled, mask_levels=mlv - 2,
                mask_feats=mask_feats, mask_boxes=boxes, mask_crops=crops)


def compare_with_gpu(path, b, ref, im_h, im_w, check_masks=True):
    """Assert that image b of a FpnRegionPath after step() equals the oracle result `ref` (bit-exact everywhere)."""
    n = int(path.n_rois[b])
    assert n == ref["rois"].shape[0], (n, ref["rois"].shape)
    for l in range(5):
        s = b * 5 + l
        k = int(path.keep_cnt[s])
        assert k == ref["scores"][l].shape[0], (l, k)
        assert np.array_equal(path.prop_scores[s, :k].cpu().numpy(), ref["scores"][l])
        assert np.array_equal(path.prop_boxes[s, :k].cpu().numpy(), ref["props"][l])
    assert np.array_equal(path.rois5[b, :n, 1:].cpu().numpy(), ref["rois"])
    assert np.array_equal(path.roi_levels[b, :n].cpu().numpy(), ref["roi_levels"])
    assert np.array_equal(path.idx_restore[b, :n].cpu().numpy(), ref["restore"])
    T = path.top_n
    bf = path.box_feats[b * T:b * T + n].float().cpu().numpy()
    assert np.abs(bf - ref["box_feats"]).max() <= 1e-4
    assert np.array_equal(bf, ref["box_feats"])
    D = int(path.det_count[b])
    assert D == ref["dets"].shape[0], (D, ref["dets"].shape)
    D = min(D, path.max_out)
    assert np.array_equal(path.dets[b, :D].cpu().numpy(), ref["dets"][:D])
    assert np.array_equal(path.det_roi[b, :D].cpu().numpy(), ref["det_roi"][:D])
    assert np.array_equal(path.m_levels[b, :D].cpu().numpy(), ref["mask_levels"][:D])
    mf = path.mask_feats[b * path.max_out:b * path.max_out + D].float().cpu().numpy()
    assert np.array_equal(mf, ref["mask_feats"][:D])
    if check_masks:
        nbytes = int(path.mask_bytes[b])
        assert nbytes <= path.crop_capacity
        crops = path.crops[b, :nbytes].cpu().numpy()
        rects = path.mask_rects[b].cpu().numpy()
        offs = path.mask_offsets[b].cpu().numpy()
        mboxes = path.mask_boxes[b].cpu().numpy()
        for d in range(D):
            bx, crop = ref["mask_boxes"][d], ref["mask_crops"][d]
            assert np.array_equal(mboxes[d], bx)
            x0, x1 = max(bx[0], 0), min(bx[2] + 1, im_w)
            y0, y1 = max(bx[1], 0), min(bx[3] + 1, im_h)
            x1, y1 = max(x1, x0), max(y1, y0)
            assert tuple(rects[d]) == (x0, y0, x1, y1)
            exp = crop[y0 - bx[1]:y1 - bx[1], x0 - bx[0]:x1 - bx[0]]
            got = crops[offs[d]:offs[d] + (x1 - x0) * (y1 - y0)].reshape(y1 - y0, x1 - x0)
            assert np.array_equal(got, exp), d
    return True


def c4_hot_path(rpn_cls, rpn_bbox, feat, cls_score, bbox_pred, sf, im_size, im_h, im_w, pre=6000, post=1000, pooled=7, sr=0,
                max_det=100, timings=None, roi_align=None):
    """BASELINE configs[1] (Faster R-CNN R-50-C4) for ONE image: rpn_cls [15,H,W], rpn_bbox [60,H,W], feat [1,C,H,W].
    Follows lib/model/detector.py:240-248, 273-284 (C4 branch) + lib/utils/result_utils.py:76-168."""
    t = time.perf_counter
    T = {} if timings is None else timings
    ra = roi_align or orc.roi_align_forward
    t0 = t()
    anchors = orc.generate_anchors(16.0)
    rois, scores = orc.generate_proposals(rpn_cls, rpn_bbox, anchors, 16.0, im_h, im_w, pre, post, 0.7)
    T["generate_proposals"] = T.get("generate_proposals", 0) + t() - t0; t0 = t()
    n = rois.shape[0]
    rois5 = np.hstack([np.zeros((n, 1), np.float32), rois])
    box_feats = ra(feat, rois5, pooled, pooled, 1.0 / 16.0, sr)
    T["roi_align_box"] = T.get("roi_align_box", 0) + t() - t0; t0 = t()
    dets, det_roi = orc.postprocess_detections(rois, sf, im_size, cls_score[:n], bbox_pred[:n], max_det=max_det)
    T["postprocess"] = T.get("postprocess", 0) + t() - t0
    return dict(rois=rois, roi_scores=scores, box_feats=box_feats, dets=dets, det_roi=det_roi)


def compare_c4_with_gpu(path, b, ref):
    """Image b of a detectorch_amd.pipeline.C4RegionPath after step() == the oracle result (bit-exact everywhere)."""
    n = int(path.n_rois[b])
    assert n == ref["rois"].shape[0], (n, ref["rois"].shape)
    assert np.array_equal(path.rois5[b, :n, 1:].cpu().numpy(), ref["rois"])
    assert np.array_equal(path.roi_scores[b, :n].cpu().numpy(), ref["roi_scores"])
    T = path.top_n
    bf = path.box_feats[b * T:b * T + n].float().cpu().numpy()
    assert np.abs(bf - ref["box_feats"]).max() <= 1e-4
    assert np.array_equal(bf, ref["box_feats"])
    D = int(path.det_count[b])
    assert D == ref["dets"].shape[0], (D, ref["dets"].shape)
    D = min(D, path.max_out)
    assert np.array_equal(path.dets[b, :D].cpu().numpy(), ref["dets"][:D])
    assert np.array_equal(path.det_roi[b, :D].cpu().numpy(), ref["det_roi"][:D])
    return True
