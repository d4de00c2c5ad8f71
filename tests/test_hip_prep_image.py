"""(f)-3 parity: dtc_prep_images (mean-subtract + bilinear resize + pad + NCHW) vs the oracle restatement, bit-exact, at small
sizes and at the BASELINE input size (a 500x833 image -> 800x1333 -> padded 800x1344), plus the size-independent
properties of the operation (constant image stays constant, scale 1 is the identity, padding is zero).  -m gpu."""
import numpy as np
import pytest
import torch

from detectorch_amd import synth

pytestmark = pytest.mark.gpu


def test_prep_images_vs_oracle_batch(oracle):
    from detectorch_amd.utils import blob as blob_utils
    rs = synth.rng(15, 0)
    ims = [rs.randint(0, 256, (h, w, 3)).astype(np.uint8) for (h, w) in [(37, 53), (60, 41), (48, 48), (20, 90)]]
    ims.append(rs.uniform(0, 255, (33, 57, 3)).astype(np.float32))
    for (target, mx, fpn) in [(64, 100, True), (64, 100, False), (24, 40, True), (200, 210, True)]:
        got, scales = blob_utils.images_to_blob(ims, target_size=target, max_size=mx, fpn_on=fpn)
        ref, rscales = oracle.prep_images(ims, target_size=target, max_size=mx, pad_stride=32 if fpn else 1)
        assert scales == rscales
        assert tuple(got.shape) == ref.shape
        assert np.array_equal(got.cpu().numpy(), ref), (target, mx, fpn)


def test_prep_full_size_and_reference_surface(oracle):
    from detectorch_amd.utils import blob as blob_utils
    rs = synth.rng(15, 1)
    im = rs.randint(0, 256, (500, 833, 3)).astype(np.uint8)
    blob, scales = blob_utils.images_to_blob([im], fpn_on=True)
    assert tuple(blob.shape) == (1, 3, 800, 1344) and abs(scales[0] - 1.6) < 1e-12          # BASELINE input shape
    ref, _ = oracle.prep_images([im])
    assert np.array_equal(blob.cpu().numpy(), ref)
    assert float(blob[0, :, :, 1333:].abs().max()) == 0.0                                    # zero padding (blob.py:45)
    # the reference's two-step surface gives the same tensor
    lst, sc = blob_utils.prep_im_for_blob(im)
    assert lst[0].shape == (800, 1333, 3) and lst[0].dtype == np.float32 and sc == scales
    two_step = blob_utils.im_list_to_blob(lst, fpn_on=True)
    assert np.array_equal(two_step, ref)
    # long-side cap: 400x1000 -> scale 1.333 (max_size / 1000), not 2.0
    _, sc2 = blob_utils.images_to_blob([np.zeros((400, 1000, 3), np.uint8)])
    assert abs(sc2[0] - 1.333) < 1e-12


def test_prep_properties():
    from detectorch_amd.utils import blob as blob_utils
    means = blob_utils.PIXEL_MEANS
    const = np.full((50, 70, 3), 200, np.uint8)
    blob, _ = blob_utils.images_to_blob([const], target_size=120, max_size=400)
    for c in range(3):      # weights sum to 1 within float rounding: a constant image stays constant
        plane = blob[0, c, :120, :168]
        assert float((plane - np.float32(200 - means[c])).abs().max()) < 1e-4
    rs = synth.rng(15, 2)
    im = rs.randint(0, 256, (64, 64, 3)).astype(np.uint8)
    blob, sc = blob_utils.images_to_blob([im], target_size=64, max_size=64)                 # scale 1: identity - mean
    assert sc[0] == 1.0
    exp = (im.astype(np.float64) - np.asarray(means)).astype(np.float32).transpose(2, 0, 1)
    assert np.array_equal(blob[0].cpu().numpy(), exp)
