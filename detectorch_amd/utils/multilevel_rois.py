"""FPN RoI -> level distribution -- same functions as the reference's lib/utils/multilevel_rois.py, computed on the GPU
(detectorch_amd/csrc/fpn.hip).  numpy in / numpy out like the reference (this is the mask-branch helper the notebooks
call between postprocess_output and model.mask_head, eval_mask_FPN.ipynb:249)."""
import numpy as np
import torch

from .. import hip


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError("detectorch_amd needs the MI355X HIP path (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _run(rois, k_min, k_max):
    rois = np.ascontiguousarray(rois, dtype=np.float32)
    n = rois.shape[0]
    box = torch.from_numpy(rois[:, -4:].copy()).to(_dev()).reshape(1, 1, max(n, 0), 4)
    if n == 0:
        return np.zeros(0, np.float32), [rois[:0] for _ in range(k_min, k_max + 1)], np.zeros(0, np.int32)
    counts = torch.tensor([[n]], dtype=torch.int32, device=box.device)
    res = hip.fpn_collect_distribute(box, None, counts, n, k_min, k_max)
    lv = res["roi_levels"][0].cpu().numpy()
    restore = res["idx_restore"][0].cpu().numpy().astype(np.int32)
    order = np.empty(n, np.int64)
    order[restore] = np.arange(n)
    cnt = res["level_counts"][0].cpu().numpy()
    per, p = [], 0
    for c in cnt:
        per.append(rois[order[p:p + int(c)], :])
        p += int(c)
    return (lv + k_min).astype(np.float32), per, restore


def map_rois_to_fpn_levels(rois, k_min, k_max, roi_canonical_scale=224, roi_canonical_level=4):
    """multilevel_rois.py:41-53 (canonical scale/level are the Detectron constants 224 / 4, fixed in the kernel)."""
    assert roi_canonical_scale == 224 and roi_canonical_level == 4
    return _run(rois, k_min, k_max)[0]


def add_multilevel_rois_for_test(blobs, name, roi_min_level=2, roi_max_level=5):
    """multilevel_rois.py:19-39 + add_multilevel_roi_blobs :56-82."""
    rois = blobs[name]
    _, per, restore = _run(rois, roi_min_level, roi_max_level)
    for i, lvl in enumerate(range(roi_min_level, roi_max_level + 1)):
        blobs[name + '_fpn' + str(lvl)] = per[i]
    blobs[name + '_idx_restore_int32'] = restore
    stacked = np.vstack(per) if len(per) else rois
    assert (stacked[restore] == np.asarray(rois, np.float32)).all()      # the reference's sanity check, :82
    return blobs
