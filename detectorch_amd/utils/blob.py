"""Network-input preparation -- the reference's lib/utils/blob.py surface for the caller side of the hot path.

    prep_im_for_blob   blob.py:62-87   mean-subtract + cv2.resize(fx=fy=im_scale, INTER_LINEAR), numpy in / numpy out
    im_list_to_blob    blob.py:27-59   zero-pad to the batch max (rounded up to the FPN stride) + HWC -> NCHW
    images_to_blob     (extension)     both in ONE launch, device in / device out: what a serving loop should call

The arithmetic runs in detectorch_amd/csrc/prep_image.hip (dtc_prep_images); there is no CPU fallback.  cv2 is not needed
(and not installed here): the resize rule is OpenCV's documented INTER_LINEAR, "parity unpinned" (DESIGN.md).
"""
import numpy as np
import torch

from .. import hip

PIXEL_MEANS = [122.7717, 115.9465, 102.9801]     # blob.py:62 (BGR)


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError("detectorch_amd.utils.blob needs the MI355X HIP path (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def images_to_blob(images, pixel_means=PIXEL_MEANS, target_size=800, max_size=1333, fpn_on=False, fpn_coarsest_stride=32):
    """list of HWC BGR images (CUDA tensors or numpy arrays, uint8 / float32) -> (blob float32 CUDA [B,3,Hb,Wb],
    im_scales list).  Equivalent to im_list_to_blob([prep_im_for_blob(im)[0][0] for im in images], fpn_on)."""
    dev = _dev()
    ts = []
    for im in images:
        t = im if torch.is_tensor(im) else torch.from_numpy(np.ascontiguousarray(im))
        if t.dtype not in (torch.uint8, torch.float32):
            t = t.to(torch.float32)
        ts.append(t.to(dev))
    blob, scales, _ = hip.prep_images(ts, pixel_means, target_size, max_size, fpn_coarsest_stride if fpn_on else 1)
    return blob, scales


def prep_im_for_blob(im, pixel_means=PIXEL_MEANS, target_sizes=[800], max_size=1333):
    """blob.py:62-87: -> (list of float32 HWC ndarrays, one per target size; list of scale factors)."""
    dev = _dev()
    t = torch.from_numpy(np.ascontiguousarray(im))
    if t.dtype not in (torch.uint8, torch.float32):
        t = t.to(torch.float32)
    t = t.to(dev)
    ims, im_scales = [], []
    for target_size in target_sizes:
        blob, scales, sizes = hip.prep_images([t], pixel_means, target_size, max_size, 1)
        oh, ow = sizes[0]
        ims.append(blob[0, :, :oh, :ow].permute(1, 2, 0).contiguous().cpu().numpy())
        im_scales.append(scales[0])
    return ims, im_scales


def im_list_to_blob(ims, fpn_on=False, fpn_coarsest_stride=32):
    """blob.py:27-59 (pure layout: pad + transpose; numpy like the reference)."""
    max_shape = np.array([im.shape for im in ims]).max(axis=0)
    if fpn_on:
        stride = float(fpn_coarsest_stride)
        max_shape[0] = int(np.ceil(max_shape[0] / stride) * stride)
        max_shape[1] = int(np.ceil(max_shape[1] / stride) * stride)
    blob = np.zeros((len(ims), max_shape[0], max_shape[1], 3), dtype=np.float32)
    for i, im in enumerate(ims):
        blob[i, 0:im.shape[0], 0:im.shape[1], :] = im
    return blob.transpose((0, 3, 1, 2))
