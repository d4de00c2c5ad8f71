"""Synthetic COCO-shaped inputs for the region-proposal hot path.

No COCO images, Detectron weights or pycocotools exist offline, so benchmark and parity inputs are drawn from fixed-seed
distributions with the shapes a 1x3x800x1333 image produces (FPN pads to 800x1344, lib/utils/blob.py:39-42 of the
reference).  numpy only; seed = 1000 * config + image_index.
"""
import numpy as np

IM_H, IM_W = 800, 1333
FPN_PAD_H, FPN_PAD_W = 800, 1344
FPN_STRIDES = (4, 8, 16, 32, 64)            # P2..P6  (detector.py:250 adds P6 by stride-2 subsampling of P5)
FPN_ROI_SCALES = (0.25, 0.125, 0.0625, 0.03125)


def fpn_level_shapes(pad_h=FPN_PAD_H, pad_w=FPN_PAD_W):
    """[(H,W)] for P2..P6: 200x336, 100x168, 50x84, 25x42, 13x21."""
    shapes = []
    for s in FPN_STRIDES[:4]:
        shapes.append((pad_h // s, pad_w // s))
    h5, w5 = shapes[-1]
    shapes.append(((h5 + 1) // 2, (w5 + 1) // 2))   # max_pool2d(k=1, stride=2)
    return shapes


def c4_shape(im_h=IM_H, im_w=IM_W):
    """res4 map of the caffe2-strided ResNet (detector.py:174-179): 50x84 for 800x1333."""
    def down(n):
        return (n + 1) // 2
    h, w = im_h, im_w
    for _ in range(4):
        h, w = down(h), down(w)
    return h, w


def dedupe_scores(s):
    """Make a float32 score array tie-free by nudging duplicates one ulp at a time (the reference's argsort/argpartition
    are unspecified on ties, generate_proposals.py:78-86, so bit-exact comparisons need distinct keys)."""
    s = np.ascontiguousarray(s, dtype=np.float32)
    flat = s.reshape(-1)
    for _ in range(64):
        _, idx, counts = np.unique(flat, return_index=True, return_counts=True)
        if flat.size == idx.size:
            break
        dup = np.ones(flat.size, bool)
        dup[idx] = False
        flat[dup] = np.nextafter(flat[dup], np.float32(0.0))
    return flat.reshape(s.shape)


def make_features(rs, shape):
    """N(0,1) ReLU'd float32 feature map [B,C,H,W]."""
    return np.maximum(rs.standard_normal(shape).astype(np.float32), 0.0)


def make_rois(rs, R, im_h=IM_H, im_w=IM_W, min_side=16.0, max_side=600.0):
    """R boxes (x1,y1,x2,y2) float32: log-uniform side, aspect exp(U(-0.7,0.7)), uniform centres, clipped to the image."""
    side = np.exp(rs.uniform(np.log(min_side), np.log(max_side), R))
    asp = np.exp(rs.uniform(-0.7, 0.7, R))
    w, h = side * np.sqrt(asp), side / np.sqrt(asp)
    cx, cy = rs.uniform(0, im_w, R), rs.uniform(0, im_h, R)
    b = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
    b[:, 0::2] = np.clip(b[:, 0::2], 0, im_w - 1)
    b[:, 1::2] = np.clip(b[:, 1::2], 0, im_h - 1)
    return b.astype(np.float32)


def harder_roi_set(batch, top_n, seed_config=3, seed_index=0):
    """The HARDER RoI population of the bench line's `roofline.harder_set` and of the real-shape parity tests: `top_n` boxes per
    image with log-uniform sides 16-600 px (a trained RPN looks like that; the bench's own RPN-on-noise proposals sit 94 % on P2),
    FPN level by area (the heuristic of multilevel_rois.py:19-39 in float64), visited in (image, level, 32-row band, x) order.
    Returns (rois5 [B*top_n,5] float32, level ids [B*top_n] int32 in 0..3, visiting order [B*top_n] int32)."""
    rs = rng(seed_config, seed_index)
    rois = np.concatenate([np.hstack([np.full((top_n, 1), b, np.float32), make_rois(rs, top_n)]) for b in range(batch)])
    area = (rois[:, 3] - rois[:, 1] + 1) * (rois[:, 4] - rois[:, 2] + 1)
    lvn = (np.clip(np.floor(4 + np.log2(np.sqrt(area) / 224 + 1e-6)), 2, 5) - 2).astype(np.int32)
    yc, xc = (rois[:, 2] + rois[:, 4]) * 0.5, (rois[:, 1] + rois[:, 3]) * 0.5
    band = (yc / (4.0 * 2.0 ** lvn * 32)).astype(np.int32)
    order = np.lexsort((xc, band, lvn, rois[:, 0])).astype(np.int32)
    return rois, lvn, order


def make_rpn_outputs(rs, A, H, W, tie_free=True):
    """(rpn_cls_prob [1,A,H,W] = sigmoid(N(-2,2)), rpn_bbox_pred [1,4A,H,W] = N(0,0.2))."""
    x = rs.standard_normal((1, A, H, W)) * 2.0 - 2.0
    p = (1.0 / (1.0 + np.exp(-x))).astype(np.float32)
    if tie_free:
        p = dedupe_scores(p)
    d = (rs.standard_normal((1, 4 * A, H, W)) * 0.2).astype(np.float32)
    return p, d


def make_head_outputs(rs, R, n_cls=81, tie_free=True):
    """(cls_score [R,n_cls] = softmax(N(0,2)), bbox_pred [R,4*n_cls] = N(0,0.1))."""
    z = rs.standard_normal((R, n_cls)) * 2.0
    z -= z.max(1, keepdims=True)
    e = np.exp(z)
    p = (e / e.sum(1, keepdims=True)).astype(np.float32)
    if tie_free:
        p = dedupe_scores(p)
    d = (rs.standard_normal((R, 4 * n_cls)) * 0.1).astype(np.float32)
    return p, d


def make_masks(rs, D, n_cls=81, M=28):
    """sigmoid(N(0,1.5)) float32 [D,n_cls,M,M] with a smooth blob so the binarised mask is not pure noise."""
    yy, xx = np.mgrid[0:M, 0:M].astype(np.float64)
    blob = -(((yy - M / 2.0) ** 2 + (xx - M / 2.0) ** 2) / (0.18 * M * M)) + 1.0
    x = rs.standard_normal((D, n_cls, M, M)) * 1.5 + 2.0 * blob
    return (1.0 / (1.0 + np.exp(-x))).astype(np.float32)


def rng(config, image_index=0):
    return np.random.RandomState(1000 * int(config) + int(image_index))
