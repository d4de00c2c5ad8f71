"""CollectAndDistributeFpnRpnProposals -- same call surface as the reference's
lib/model/collect_and_distribute_fpn_rpn_proposals.py:35-81, computed by one HIP workgroup per image
(detectorch_amd/csrc/fpn.hip) instead of torch.sort + D2H + numpy.

forward(roi_list, roi_score_list) -> (list of per-level roi tensors [R_l,4] for levels k_min..k_max, idx_restore ndarray)
exactly like :81/:128 (idx_restore is a numpy array used to index torch tensors, lib/model/detector.py:269-270).
"""
from math import log2

import numpy as np
import torch

from .. import hip


class CollectAndDistributeFpnRpnProposals(torch.nn.Module):
    def __init__(self, spatial_scales, train=False):
        super(CollectAndDistributeFpnRpnProposals, self).__init__()
        self._train = train
        self.rpn_levels = [int(log2(1 / s)) for s in spatial_scales]      # :39
        self.rpn_min_level = self.rpn_levels[0]
        self.rpn_max_level = self.rpn_levels[-1]

    def forward(self, roi_list, roi_score_list):
        post_nms_topN = 2000 if self._train else 1000                    # :86
        res = collect_and_distribute(roi_list, roi_score_list, post_nms_topN, self.rpn_min_level, self.rpn_max_level)
        n = int(res["n_out"][0].item())
        counts = res["level_counts"][0].cpu().numpy()
        by_level = res["rois_by_level"][0]
        distr, p = [], 0
        for c in counts:
            distr.append(by_level[p:p + int(c), :])
            p += int(c)
        return distr, res["idx_restore"][0, :n].cpu().numpy().astype(np.int64)


def collect_and_distribute(roi_list, roi_score_list, post_nms_topN, lvl_min, lvl_max):
    """Single image: pack the per-level lists into the batched layout of dtc_fpn_collect_distribute."""
    L = len(roi_list)
    dev = roi_list[0].device
    P = max(1, max(int(r.shape[0]) for r in roi_list))
    boxes = torch.zeros((1, L, P, 4), dtype=torch.float32, device=dev)
    scores = torch.zeros((1, L, P), dtype=torch.float32, device=dev)
    counts = torch.tensor([[int(r.shape[0]) for r in roi_list]], dtype=torch.int32, device=dev)
    for l in range(L):
        n = int(roi_list[l].shape[0])
        if n:
            boxes[0, l, :n] = roi_list[l].reshape(n, -1)[:, -4:]
            scores[0, l, :n] = roi_score_list[l].reshape(-1)
    return hip.fpn_collect_distribute(boxes, scores, counts, post_nms_topN, lvl_min, lvl_max)
