"""Box utilities -- the subset of the reference's lib/utils/boxes.py that sits on the region-proposal hot path, with the
same names and numpy-in / numpy-out conventions, computed by the HIP kernels (detectorch_amd/csrc).

    nms              boxes.py:332-336   -> dtc_nms            (cython_nms.pyx:37-87)
    soft_nms         boxes.py:339-356   -> dtc_soft_nms       (cython_nms.pyx:98-203)
    bbox_transform   boxes.py:168-208   -> dtc_bbox_transform
    clip_tiled_boxes boxes.py:150-165
    expand_boxes     boxes.py:245-261 ; boxes_area boxes.py:75-81  (trivial host numpy, identical arithmetic)

There is no CPU fallback for the kernels: without the GPU library these raise.
"""
import numpy as np
import torch

from .. import hip

cfg_BBOX_XFORM_CLIP = 4.135166556742356   # boxes.py:73


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError("detectorch_amd.utils.boxes needs the MI355X HIP path (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def nms(dets, thresh):
    """Apply classic DPM-style greedy NMS (boxes.py:332).  dets: ndarray [N,5] -> ndarray int64 of kept indices."""
    if dets.shape[0] == 0:
        return []                                              # boxes.py:334-335
    d = torch.from_numpy(np.ascontiguousarray(dets, dtype=np.float32)).to(_dev())
    return hip.nms(d, thresh).cpu().numpy()


def soft_nms(dets, sigma=0.5, overlap_thresh=0.3, score_thresh=0.001, method='linear'):
    """Apply the soft NMS algorithm (boxes.py:339-356) -> (dets', keep)."""
    if dets.shape[0] == 0:
        return dets, []
    methods = {'hard': 0, 'linear': 1, 'gaussian': 2}
    assert method in methods, 'Unknown soft_nms method: {}'.format(method)
    d = torch.from_numpy(np.ascontiguousarray(dets, dtype=np.float32)).to(_dev())
    out, keep = hip.soft_nms(d, np.float32(sigma), np.float32(overlap_thresh), np.float32(score_thresh), methods[method])
    return out.cpu().numpy(), keep.cpu().numpy()


def bbox_transform(boxes, deltas, weights=(1.0, 1.0, 1.0, 1.0)):
    """Forward transform boxes + regression deltas -> predicted boxes (boxes.py:168-208)."""
    if boxes.shape[0] == 0:
        return np.zeros((0, deltas.shape[1]), dtype=deltas.dtype)
    dev = _dev()
    b = torch.from_numpy(np.ascontiguousarray(boxes, dtype=np.float32)).to(dev)
    d = torch.from_numpy(np.ascontiguousarray(deltas, dtype=np.float32)).to(dev)
    return hip.bbox_transform(b, d, weights).cpu().numpy()


def clip_tiled_boxes(boxes, im_shape):
    """Clip boxes to image boundaries (boxes.py:150-165); exact min/max, done in place on the host array like the reference."""
    assert boxes.shape[1] % 4 == 0
    boxes[:, 0::4] = np.maximum(np.minimum(boxes[:, 0::4], im_shape[1] - 1), 0)
    boxes[:, 1::4] = np.maximum(np.minimum(boxes[:, 1::4], im_shape[0] - 1), 0)
    boxes[:, 2::4] = np.maximum(np.minimum(boxes[:, 2::4], im_shape[1] - 1), 0)
    boxes[:, 3::4] = np.maximum(np.minimum(boxes[:, 3::4], im_shape[0] - 1), 0)
    return boxes


def boxes_area(boxes):
    w = (boxes[:, 2] - boxes[:, 0] + 1)
    h = (boxes[:, 3] - boxes[:, 1] + 1)
    areas = w * h
    assert np.all(areas >= 0), 'Negative areas founds'
    return areas


def expand_boxes(boxes, scale):
    w_half = (boxes[:, 2] - boxes[:, 0]) * .5
    h_half = (boxes[:, 3] - boxes[:, 1]) * .5
    x_c = (boxes[:, 2] + boxes[:, 0]) * .5
    y_c = (boxes[:, 3] + boxes[:, 1]) * .5
    w_half *= scale
    h_half *= scale
    boxes_exp = np.zeros(boxes.shape)
    boxes_exp[:, 0] = x_c - w_half
    boxes_exp[:, 2] = x_c + w_half
    boxes_exp[:, 1] = y_c - h_half
    boxes_exp[:, 3] = y_c + h_half
    return boxes_exp
