"""RoIAlign -- same call surface as the reference's lib/model/roi_align.py (RoIAlignFunction :23, RoIAlign :150,
preprocess_rois :172), backed by the hand-written gfx950 kernel (detectorch_amd/csrc/roi_align.hip) instead of the
torch-0.4 JIT / torch-0.3 cffi CUDA extensions the reference selects at :8-20.

Differences a caller can observe:
  * GPU only: CPU tensors raise (the reference's CPU branch :67-83 is what oracle/ restates as the checker);
  * inference only: backward raises NotImplementedError (the reference's backward :93-145 is training code, out of the
    hot-path scope -- README.md:3);
  * features may be NCHW-contiguous or channels_last, float32 or float16.
"""
import torch
from torch.autograd import Function
from torch.nn.modules.module import Module

from .. import hip


class RoIAlignFunction(Function):
    @staticmethod
    def forward(ctx, features, rois, pooled_height, pooled_width, spatial_scale, sampling_ratio):
        # reference: roi_align.py:33-89
        if features.is_cuda != rois.is_cuda:
            raise TypeError('features and rois should be on same device (CPU or GPU)')   # :43-44
        ctx.mark_non_differentiable()
        return hip.roi_align_forward(features, float(spatial_scale), rois, int(pooled_height), int(pooled_width),
                                     int(sampling_ratio))

    @staticmethod
    def backward(ctx, grad_output):
        raise NotImplementedError("detectorch_amd.RoIAlign is inference-only (reference backward: roi_align.py:93-145)")


class RoIAlign(Module):
    def __init__(self, pooled_height, pooled_width, spatial_scale, sampling_ratio=0):
        super(RoIAlign, self).__init__()
        self.pooled_height = int(pooled_height)
        self.pooled_width = int(pooled_width)
        self.spatial_scale = float(spatial_scale)
        self.sampling_ratio = int(sampling_ratio)

    def forward(self, features, rois):
        rois = preprocess_rois(rois)
        return RoIAlignFunction.apply(features, rois, self.pooled_height, self.pooled_width, self.spatial_scale,
                                      self.sampling_ratio)


def preprocess_rois(rois):
    """reference: roi_align.py:172-187 -- list -> cat; [1,R,k] -> [R,k]; 4 columns -> prepend a zero batch column."""
    if isinstance(rois, list):
        rois = torch.cat(tuple(rois), 0)
    if torch.is_tensor(rois):
        if rois.dim() == 3:
            if rois.size(0) == 1:
                rois = rois.squeeze(0)
            else:
                raise ValueError("rois has wrong size")
        if rois.size(1) == 4:
            zeros = torch.zeros((rois.size(0), 1), dtype=rois.dtype, device=rois.device)
            rois = torch.cat((zeros, rois), 1).contiguous()
    return rois
