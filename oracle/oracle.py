"""numpy wrappers over oracle/liboracle.so (the plain-C restatement in oracle.c).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never
by detectorch_amd/.  Function names mirror the reference functions they restate (see oracle.c for file:line).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(HERE, "liboracle.so")
    src = os.path.join(HERE, "oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        p, i, f, d = C.c_void_p, C.c_int, C.c_float, C.c_double
        _LIB.orc_roi_align_forward.argtypes = [p, p, i, i, f, i, i, i, i, i, i, p]
        _LIB.orc_roi_align_forward.restype = None
        _LIB.orc_generate_anchors.argtypes = [d, p, i, p, i, p]
        _LIB.orc_generate_anchors.restype = None
        _LIB.orc_bbox_transform.argtypes = [p, p, i, i, f, f, f, f, p]
        _LIB.orc_bbox_transform.restype = None
        _LIB.orc_clip_tiled_boxes.argtypes = [p, i, f, f]
        _LIB.orc_clip_tiled_boxes.restype = None
        _LIB.orc_nms.argtypes = [p, i, f, i, p]
        _LIB.orc_nms.restype = i
        _LIB.orc_soft_nms.argtypes = [p, i, f, f, f, C.c_uint, p]
        _LIB.orc_soft_nms.restype = i
        _LIB.orc_generate_proposals.argtypes = [p, p, i, i, i, p, d, f, f, f, i, i, f, p, p, p, p, p]
        _LIB.orc_generate_proposals.restype = i
        _LIB.orc_map_rois_to_fpn_levels.argtypes = [p, i, i, i, p]
        _LIB.orc_map_rois_to_fpn_levels.restype = None
        _LIB.orc_collect.argtypes = [p, p, i, i, p, p, p]
        _LIB.orc_collect.restype = i
        _LIB.orc_distribute.argtypes = [p, i, i, i, p, p, p]
        _LIB.orc_distribute.restype = None
        _LIB.orc_postprocess_detections.argtypes = [p, i, f, p, p, i, f, f, f, f, f, f, f, f, i, p, p]
        _LIB.orc_postprocess_detections.restype = i
        _LIB.orc_expand_box_int.argtypes = [p, i, p]
        _LIB.orc_expand_box_int.restype = None
        _LIB.orc_mask_resize_binarize.argtypes = [p, i, p, f, p, p]
        _LIB.orc_mask_resize_binarize.restype = i
        _LIB.orc_bbox_overlaps.argtypes = [p, i, p, i, p]
        _LIB.orc_bbox_overlaps.restype = None
        _LIB.orc_box_voting.argtypes = [p, i, p, i, f, p]
        _LIB.orc_box_voting.restype = i
        _LIB.orc_prep_scale.argtypes = [i, i, i, i]
        _LIB.orc_prep_scale.restype = C.c_double
        _LIB.orc_prep_image.argtypes = [p, i, i, i, p, C.c_double, i, i, p, i, i]
        _LIB.orc_prep_image.restype = None
        _LIB.orc_rle_runs.argtypes = [p, i, i, p]
        _LIB.orc_rle_runs.restype = C.c_longlong
        _LIB.orc_rle_string.argtypes = [p, C.c_longlong, p]
        _LIB.orc_rle_string.restype = C.c_longlong
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def roi_align_forward(features, rois, pooled_h, pooled_w, spatial_scale, sampling_ratio):
    features, rois = _f32(features), _f32(rois)
    _, Cn, H, W = features.shape
    R, cols = rois.shape if rois.size else (0, 5)
    out = np.zeros((R, Cn, pooled_h, pooled_w), dtype=np.float32)
    lib().orc_roi_align_forward(features.ctypes.data, rois.ctypes.data, R, cols, spatial_scale, Cn, H, W, pooled_h,
                                pooled_w, sampling_ratio, out.ctypes.data)
    return out


def generate_anchors(stride=16, sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1, 2)):
    s = np.asarray(sizes, dtype=np.float64)
    r = np.asarray(aspect_ratios, dtype=np.float64)
    out = np.zeros((len(r) * len(s), 4), dtype=np.float64)
    lib().orc_generate_anchors(float(stride), s.ctypes.data, len(s), r.ctypes.data, len(r), out.ctypes.data)
    return out


def bbox_transform(boxes, deltas, weights=(1.0, 1.0, 1.0, 1.0)):
    boxes, deltas = _f32(boxes), _f32(deltas)
    n = boxes.shape[0]
    ncls = deltas.shape[1] // 4
    out = np.zeros_like(deltas)
    lib().orc_bbox_transform(boxes.ctypes.data, deltas.ctypes.data, n, ncls, *[float(w) for w in weights],
                             out.ctypes.data)
    return out


def clip_tiled_boxes(boxes, im_h, im_w):
    boxes = _f32(boxes).copy()
    lib().orc_clip_tiled_boxes(boxes.ctypes.data, boxes.size // 4, float(im_h), float(im_w))
    return boxes


def nms(dets, thresh, max_keep=0):
    dets = _f32(dets)
    n = dets.shape[0]
    keep = np.zeros(max(n, 1), dtype=np.int64)
    k = lib().orc_nms(dets.ctypes.data, n, float(np.float32(thresh)), int(max_keep), keep.ctypes.data)
    return keep[:k].copy()


def soft_nms(dets, sigma=0.5, overlap_thresh=0.3, score_thresh=0.001, method="linear"):
    methods = {"hard": 0, "linear": 1, "gaussian": 2}
    boxes = _f32(dets).copy()
    n = boxes.shape[0]
    inds = np.zeros(max(n, 1), dtype=np.int64)
    k = lib().orc_soft_nms(boxes.ctypes.data, n, float(np.float32(sigma)), float(np.float32(overlap_thresh)),
                           float(np.float32(score_thresh)), methods[method], inds.ctypes.data)
    return boxes[:k].copy(), inds[:k].copy()


def rpn_sigmoid(logits):
    """rpn_cls_probs = sigmoid(rpn_cls_logits) (reference: lib/model/detector.py:125, F.sigmoid on float32).  Restated as the
    correctly-rounded value (evaluate in float64, round once) so that the checker does not depend on a libm's float32
    ulp behaviour; torch's own float32 sigmoid is within 2 ulp of it (checked in tests/test_oracle_golden.py)."""
    x = np.asarray(logits, np.float32).astype(np.float64)
    return (1.0 / (1.0 + np.exp(-x))).astype(np.float32)


def softmax_rows(logits):
    """F.softmax(cls_score) of lib/model/detector.py:281 restated as the correctly-rounded value: exp and the sum in float64,
    one rounding to float32 at the end.  The sum is taken in the order the HIP kernel uses (element j into slot j % 64 in
    increasing j, then a 64 -> 1 halving tree) so that the two agree bit for bit; torch's own float32 softmax agrees to rel 1e-6
    (tests/test_oracle_golden.py)."""
    l = np.asarray(logits, np.float32)
    m = l.max(axis=-1, keepdims=True).astype(np.float64)
    e = np.exp(l.astype(np.float64) - m)
    n = l.shape[-1]
    s = np.zeros(l.shape[:-1] + (64,), np.float64)
    for j in range(n):
        s[..., j % 64] += e[..., j]
    for off in (32, 16, 8, 4, 2, 1):
        s = s[..., :off] + s[..., off:2 * off]
    return (e / s).astype(np.float32)


def generate_proposals(scores, deltas, anchors, feat_stride, im_h, im_w, pre_nms_top_n, post_nms_top_n, nms_thresh,
                       min_size_scaled=0.0, return_pre_nms=False):
    """scores [A,H,W], deltas [4A,H,W] -> (boxes [k,4], scores [k])."""
    scores, deltas = _f32(scores), _f32(deltas)
    anchors = np.ascontiguousarray(anchors, dtype=np.float64)
    A, H, W = scores.shape
    N = A * H * W
    K = N if (pre_nms_top_n <= 0 or pre_nms_top_n >= N) else pre_nms_top_n
    cap = max(K if (post_nms_top_n <= 0 or nms_thresh <= 0) else min(K, post_nms_top_n), 1)
    ob = np.zeros((cap, 4), np.float32)
    os_ = np.zeros((cap,), np.float32)
    pb = np.zeros((K, 4), np.float32)
    ps = np.zeros((K,), np.float32)
    pn = C.c_int(0)
    k = lib().orc_generate_proposals(scores.ctypes.data, deltas.ctypes.data, A, H, W, anchors.ctypes.data,
                                     float(feat_stride), float(im_h), float(im_w), float(min_size_scaled),
                                     int(pre_nms_top_n), int(post_nms_top_n), float(np.float32(nms_thresh)),
                                     ob.ctypes.data, os_.ctypes.data, pb.ctypes.data, ps.ctypes.data, C.byref(pn))
    if return_pre_nms:
        return ob[:k].copy(), os_[:k].copy(), pb[:pn.value].copy(), ps[:pn.value].copy()
    return ob[:k].copy(), os_[:k].copy()


def map_rois_to_fpn_levels(rois, k_min, k_max):
    rois = _f32(rois)
    n = rois.shape[0]
    lv = np.zeros(max(n, 1), np.int32)
    lib().orc_map_rois_to_fpn_levels(rois.ctypes.data, n, k_min, k_max, lv.ctypes.data)
    return lv[:n].copy()


def collect(rois_cat, scores_cat, post_nms_topN):
    rois_cat, scores_cat = _f32(rois_cat), _f32(scores_cat).reshape(-1)
    n = rois_cat.shape[0]
    m = min(n, post_nms_topN)
    out = np.zeros((max(m, 1), 4), np.float32)
    osc = np.zeros(max(m, 1), np.float32)
    src = np.zeros(max(m, 1), np.int64)
    k = lib().orc_collect(rois_cat.ctypes.data, scores_cat.ctypes.data, n, post_nms_topN, out.ctypes.data,
                          osc.ctypes.data, src.ctypes.data)
    return out[:k].copy(), osc[:k].copy(), src[:k].copy()


def distribute(rois, k_min, k_max):
    """-> (list of per-level roi arrays, idx_restore, lvls)"""
    rois = _f32(rois)
    n = rois.shape[0]
    lv = map_rois_to_fpn_levels(rois, k_min, k_max)
    counts = np.zeros(k_max - k_min + 1, np.int32)
    order = np.zeros(max(n, 1), np.int64)
    restore = np.zeros(max(n, 1), np.int64)
    lib().orc_distribute(lv.ctypes.data, n, k_min, k_max, counts.ctypes.data, order.ctypes.data, restore.ctypes.data)
    outs, p = [], 0
    for c in counts:
        outs.append(rois[order[p:p + c]])
        p += c
    return outs, restore[:n].copy(), lv


def postprocess_detections(rois, scaling_factor, im_size, cls_scores, bbox_deltas, weights=(10.0, 10.0, 5.0, 5.0),
                           score_thresh=0.05, nms_thresh=0.5, max_det=100):
    """-> (dets [D,6] (x1,y1,x2,y2,score,class), roi_index [D])"""
    rois, cls_scores, bbox_deltas = _f32(rois), _f32(cls_scores), _f32(bbox_deltas)
    R, ncls = cls_scores.shape
    cap = max(R * (ncls - 1), 1)
    dets = np.zeros((cap, 6), np.float32)
    src = np.zeros(cap, np.int32)
    D = lib().orc_postprocess_detections(rois.ctypes.data, R, float(np.float32(scaling_factor)),
                                         cls_scores.ctypes.data, bbox_deltas.ctypes.data, ncls, float(im_size[0]),
                                         float(im_size[1]), *[float(w) for w in weights],
                                         float(np.float32(score_thresh)), float(np.float32(nms_thresh)), int(max_det),
                                         dets.ctypes.data, src.ctypes.data)
    return dets[:D].copy(), src[:D].copy()


def expand_box_int(ref_box, M):
    ref_box = _f32(ref_box)
    out = np.zeros(4, np.int32)
    lib().orc_expand_box_int(ref_box.ctypes.data, int(M), out.ctypes.data)
    return out


def mask_resize_binarize(mask, ref_box, thresh=0.5):
    """mask [M,M] -> (box int32[4], crop uint8 [h,w])"""
    mask, ref_box = _f32(mask), _f32(ref_box)
    M = mask.shape[0]
    box = np.zeros(4, np.int32)
    n = lib().orc_mask_resize_binarize(mask.ctypes.data, M, ref_box.ctypes.data, float(np.float32(thresh)),
                                       box.ctypes.data, None)
    crop = np.zeros(n, np.uint8)
    lib().orc_mask_resize_binarize(mask.ctypes.data, M, ref_box.ctypes.data, float(np.float32(thresh)),
                                   box.ctypes.data, crop.ctypes.data)
    w, h = max(box[2] - box[0] + 1, 1), max(box[3] - box[1] + 1, 1)
    return box, crop.reshape(h, w)


def rle_encode(mask):
    """mask [h,w] uint8 -> (runs uint32 [n], counts str): the COCO RLE of result_utils.py:217-220 (pycocotools restated)."""
    mask = np.ascontiguousarray(mask, np.uint8)
    h, w = mask.shape
    n = lib().orc_rle_runs(mask.ctypes.data, h, w, None)
    runs = np.zeros(n, np.uint32)
    lib().orc_rle_runs(mask.ctypes.data, h, w, runs.ctypes.data)
    buf = np.zeros(7 * n, np.uint8)
    m = lib().orc_rle_string(runs.ctypes.data, n, buf.ctypes.data)
    return runs, buf[:m].tobytes().decode("ascii")


def bbox_overlaps(boxes, query_boxes):
    """cython_bbox.bbox_overlaps (cython_bbox.pyx:32-72): [N,4] x [K,4] -> [N,K] float32."""
    b, q = _f32(boxes), _f32(query_boxes)
    out = np.zeros((b.shape[0], q.shape[0]), np.float32)
    lib().orc_bbox_overlaps(b.ctypes.data, b.shape[0], q.ctypes.data, q.shape[0], out.ctypes.data)
    return out


def box_voting(top_dets, all_dets, thresh):
    """boxes.py:280-329 with scoring_method='ID': [T,5], [A,5] -> [T,5]."""
    t, a = _f32(top_dets), _f32(all_dets)
    out = np.zeros_like(t)
    lib().orc_box_voting(t.ctypes.data, t.shape[0], a.ctypes.data, a.shape[0], float(np.float32(thresh)), out.ctypes.data)
    return out


def prep_images(images, pixel_means=(122.7717, 115.9465, 102.9801), target_size=800, max_size=1333, pad_stride=32):
    """blob.py:62-87 + :27-59 for a list of HWC BGR images (uint8 or float32) -> (blob float32 [B,3,Hb,Wb], scales list)."""
    means = np.asarray(pixel_means, np.float64)
    scales, sizes = [], []
    for im in images:
        h, w = im.shape[:2]
        s = lib().orc_prep_scale(h, w, int(target_size), int(max_size))
        scales.append(s)
        sizes.append((max(int(np.round(h * s)), 1), max(int(np.round(w * s)), 1)))      # np.round == rint: half to even
    Hb, Wb = max(a for a, _ in sizes), max(b for _, b in sizes)
    if pad_stride > 1:
        Hb = int(np.ceil(Hb / float(pad_stride)) * pad_stride); Wb = int(np.ceil(Wb / float(pad_stride)) * pad_stride)
    blob = np.zeros((len(images), 3, Hb, Wb), np.float32)
    for b, im in enumerate(images):
        u8 = im.dtype == np.uint8
        src = np.ascontiguousarray(im if u8 else im.astype(np.float32))
        lib().orc_prep_image(src.ctypes.data, 1 if u8 else 0, src.shape[0], src.shape[1], means.ctypes.data, scales[b],
                             sizes[b][0], sizes[b][1], blob[b].ctypes.data, Hb, Wb)
    return blob, scales
