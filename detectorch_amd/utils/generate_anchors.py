"""Base anchor enumeration -- same function as the reference's lib/utils/generate_anchors.py:54-122 (Detectron).

Tiny (A <= 15 anchors), computed once on the host in float64 and uploaded as kernel constants; the per-position shifted
anchors of generate_proposals.py:124-149 are never materialised (the HIP kernel derives anchor = base[a] + shift(h, w)).
Known-answer: generate_anchors.py:26-51 lists the stride-16 / scales 8,16,32 table in 1-based pixels; this returns it
minus 1 (0-based), exactly like the reference code does.
"""
import numpy as np


def generate_anchors(stride=16, sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1, 2)):
    return _generate_anchors(stride, np.array(sizes, dtype=np.float64) / stride,
                             np.array(aspect_ratios, dtype=np.float64))


def _whctrs(anchor):
    w = anchor[2] - anchor[0] + 1
    h = anchor[3] - anchor[1] + 1
    return w, h, anchor[0] + 0.5 * (w - 1), anchor[1] + 0.5 * (h - 1)


def _mkanchors(ws, hs, x_ctr, y_ctr):
    ws, hs = ws[:, np.newaxis], hs[:, np.newaxis]
    return np.hstack((x_ctr - 0.5 * (ws - 1), y_ctr - 0.5 * (hs - 1), x_ctr + 0.5 * (ws - 1), y_ctr + 0.5 * (hs - 1)))


def _generate_anchors(base_size, scales, aspect_ratios):
    anchor = np.array([1, 1, base_size, base_size], dtype=np.float64) - 1
    w, h, x_ctr, y_ctr = _whctrs(anchor)
    ws = np.round(np.sqrt(w * h / aspect_ratios))          # np.round: half-to-even, as the reference
    hs = np.round(ws * aspect_ratios)
    ratio_anchors = _mkanchors(ws, hs, x_ctr, y_ctr)
    out = []
    for i in range(ratio_anchors.shape[0]):
        w, h, x_ctr, y_ctr = _whctrs(ratio_anchors[i, :])
        out.append(_mkanchors(w * scales, h * scales, x_ctr, y_ctr))
    return np.vstack(out)
