"""GenerateProposals -- same constructor / forward signature as the reference's lib/model/generate_proposals.py:13-31,
computed entirely on the MI355X (detectorch_amd/csrc/proposals.hip + nms.hip): no numpy, no D2H of scores/deltas, no
host argpartition, no Cython NMS.

forward(rpn_cls_probs[1,A,H,W], rpn_bbox_pred[1,4A,H,W], im_height, im_width, scaling_factor, spatial_scale=None)
    -> (proposals [k,4], scores [k,1])  on the input device, like generate_proposals.py:122.
Tie rule where the reference's numpy sorts are unspecified (:78-86): score descending, then (h,w,a) index ascending.
`forward_batched` is the MI355X-first entry: every image and every FPN level in one call.
"""
import torch

from .. import hip
from ..utils.generate_anchors import generate_anchors


class GenerateProposals(torch.nn.Module):
    def __init__(self, spatial_scale=0.0625, train=False, rpn_pre_nms_top_n=None, rpn_post_nms_top_n=None,
                 rpn_nms_thresh=None, rpn_min_size=0, anchor_sizes=(32, 64, 128, 256, 512),
                 anchor_aspect_ratios=(0.5, 1, 2)):
        super(GenerateProposals, self).__init__()
        # generate_proposals.py:22-29
        self._anchors = generate_anchors(sizes=anchor_sizes, aspect_ratios=anchor_aspect_ratios,
                                         stride=1. / spatial_scale)
        self._num_anchors = self._anchors.shape[0]
        self._spatial_scale = spatial_scale
        self._train = train
        self.rpn_pre_nms_top_n = rpn_pre_nms_top_n if rpn_pre_nms_top_n is not None else (12000 if train else 6000)
        self.rpn_post_nms_top_n = rpn_post_nms_top_n if rpn_post_nms_top_n is not None else (2000 if train else 1000)
        self.rpn_nms_thresh = rpn_nms_thresh if rpn_nms_thresh is not None else 0.7
        self.rpn_min_size = rpn_min_size if rpn_min_size is not None else 0

    def forward(self, rpn_cls_probs, rpn_bbox_pred, im_height, im_width, scaling_factor, spatial_scale=None, *,
                scores_are_logits=False):
        """Reference signature (generate_proposals.py:31).  Extension (keyword-only): scores_are_logits=True takes the
        PRE-sigmoid rpn_cls_logits and returns what the reference returns for sigmoid(logits) -- the sigmoid of
        detector.py:125 is folded into the top-k kernel."""
        if spatial_scale is None:
            spatial_scale = self._spatial_scale
        if rpn_cls_probs.shape[0] != 1:
            raise ValueError("GenerateProposals.forward is batch-1 like the reference (squeeze(0) at :64,72); "
                             "use detectorch_amd.hip.generate_proposals for batches")
        sf = float(scaling_factor.reshape(-1)[0]) if torch.is_tensor(scaling_factor) else float(scaling_factor)
        boxes, scores, counts, _, _, _ = hip.generate_proposals(
            [rpn_cls_probs], [rpn_bbox_pred], [self._anchors], [1. / spatial_scale], im_height, im_width,
            [self.rpn_pre_nms_top_n], self.rpn_post_nms_top_n, self.rpn_nms_thresh,
            min_size_scaled=self.rpn_min_size * sf, scores_are_logits=scores_are_logits)
        k = int(counts.reshape(-1)[0].item())            # variable-length return value => one sync, as in the reference
        return boxes[0, 0, :k, :], scores[0, 0, :k].unsqueeze(1)
