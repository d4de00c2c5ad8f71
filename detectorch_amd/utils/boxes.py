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
