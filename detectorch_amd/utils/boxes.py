"""Box utilities -- the subset of the reference's lib/utils/boxes.py that sits on the region-proposal hot path, with the
same names and numpy-in / numpy-out conventions, computed by the HIP kernels (detectorch_amd/csrc).

    nms              boxes.py:332-336   -> dtc_nms            (cython_nms.pyx:37-87)
    soft_nms         boxes.py:339-356   -> dtc_soft_nms       (cython_nms.pyx:98-203)
    bbox_transform   boxes.py:168-208   -> dtc_bbox_transform
    bbox_overlaps    boxes.py:53-69 (cython_bbox.pyx:32-72) -> dtc_bbox_overlaps
    box_voting       boxes.py:280-329   -> dtc_box_voting (+ host statistics for the non-'ID' scoring methods)
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


def nms(dets, thresh, tie_break='index_asc'):
    """Apply classic DPM-style greedy NMS (boxes.py:332).  dets: ndarray [N,5] -> ndarray int64 of kept indices.

    tie_break: visiting order of boxes with EQUAL scores.  The reference visits `scores.argsort()[::-1]`
    (cython_nms.pyx:45): the reverse of an ascending argsort, i.e. equal scores in DESCENDING index order whenever numpy's
    sort happens to be stable (it is unspecified for the default kind).  'index_asc' (default) is this library's canonical
    rule everywhere (score desc, index asc); 'index_desc' reproduces the stable-argsort reading of the reference -- done on
    the host by running the same kernel on the row-reversed array and mapping the kept indices back."""
    if dets.shape[0] == 0:
        return []                                              # boxes.py:334-335
    assert tie_break in ('index_asc', 'index_desc')
    a = np.ascontiguousarray(dets, dtype=np.float32)
    if tie_break == 'index_desc':
        a = np.ascontiguousarray(a[::-1])
    keep = hip.nms(torch.from_numpy(a).to(_dev()), thresh).cpu().numpy()
    if tie_break == 'index_desc':
        keep = np.sort(a.shape[0] - 1 - keep)
    return keep


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


def bbox_overlaps(boxes, query_boxes):
    """IoU matrix [N,K] float32 of boxes [N,4] vs query_boxes [K,4] (cython_bbox.bbox_overlaps, cython_bbox.pyx:32-72)."""
    b = np.ascontiguousarray(boxes, dtype=np.float32)
    q = np.ascontiguousarray(query_boxes, dtype=np.float32)
    if b.shape[0] == 0 or q.shape[0] == 0:
        return np.zeros((b.shape[0], q.shape[0]), dtype=np.float32)
    dev = _dev()
    return hip.bbox_overlaps(torch.from_numpy(b).to(dev), torch.from_numpy(q).to(dev)).cpu().numpy()


def box_voting(top_dets, all_dets, thresh, scoring_method='ID', beta=1.0):
    """Bounding-box voting (boxes.py:280-329): every row of top_dets [N,5] gets its box replaced by the score-weighted
    mean of the all_dets [M,5] that overlap it by IoU >= thresh (device: dtc_box_voting).  scoring_method other than 'ID'
    also rewrites the score column from the voters' scores -- small host statistics over the device IoU matrix."""
    methods = ('ID', 'TEMP_AVG', 'AVG', 'IOU_AVG', 'GENERALIZED_AVG', 'QUASI_SUM')
    if scoring_method not in methods:
        raise NotImplementedError('Unknown scoring method {}'.format(scoring_method))       # boxes.py:324-327
    top = np.ascontiguousarray(top_dets, dtype=np.float32)
    alld = np.ascontiguousarray(all_dets, dtype=np.float32)
    if top.shape[0] == 0:
        return top.copy()
    dev = _dev()
    t_top, t_all = torch.from_numpy(top).to(dev), torch.from_numpy(alld).to(dev)
    voted, n_voters = hip.box_voting(t_top, t_all, np.float32(thresh))
    if int(n_voters.min().item()) == 0:
        raise ZeroDivisionError("Weights sum to zero, can't be normalized")                 # what np.average raises at :295
    out = voted.cpu().numpy()
    if scoring_method == 'ID':
        return out
    iou = hip.bbox_overlaps(t_top[:, :4].contiguous(), t_all[:, :4].contiguous()).cpu().numpy()
    scores_all = alld[:, 4]
    for k in range(out.shape[0]):
        sel = np.where(iou[k] >= thresh)[0]
        ws = scores_all[sel]
        if scoring_method == 'AVG':                                                          # :311-313
            out[k, 4] = ws.mean()
        elif scoring_method == 'IOU_AVG':                                                    # :314-318
            out[k, 4] = np.average(ws, weights=iou[k, sel])
        elif scoring_method == 'GENERALIZED_AVG':                                            # :319-321
            out[k, 4] = np.mean(ws ** beta) ** (1.0 / beta)
        elif scoring_method == 'QUASI_SUM':                                                  # :322-323
            out[k, 4] = ws.sum() / float(len(ws)) ** beta
        else:                                                                                # 'TEMP_AVG' :300-310
            two = np.vstack((ws, 1.0 - ws))
            logit = np.log(two / np.max(two, axis=0))
            soft = np.exp(logit / beta)
            out[k, 4] = (soft / np.sum(soft, axis=0))[0].mean()
    return out


def xyxy_to_xywh(xyxy):
    """lib/utils/boxes.py:110-123: [x1 y1 x2 y2] -> [x1 y1 w h] (w = x2 - x1 + 1)."""
    if isinstance(xyxy, (list, tuple)):
        assert len(xyxy) == 4
        x1, y1 = xyxy[0], xyxy[1]
        return (x1, y1, xyxy[2] - x1 + 1, xyxy[3] - y1 + 1)
    if isinstance(xyxy, np.ndarray):
        return np.hstack((xyxy[:, 0:2], xyxy[:, 2:4] - xyxy[:, 0:2] + 1))
    raise TypeError('Argument xyxy must be a list, tuple, or numpy array.')
