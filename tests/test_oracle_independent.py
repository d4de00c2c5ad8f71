"""Independent cross-checks of the oracle legs whose arithmetic lives in third-party code that is absent here (OpenCV's
cv2.resize at lib/utils/result_utils.py:202 and lib/utils/blob.py:82-85; pycocotools' RLE at result_utils.py:217-220).

These do NOT pin the oracle to the reference's own dependency (that stays "parity unpinned"): they pin
it to a SECOND, independently written implementation of the same published rule, so that "HIP == oracle" is more than a
self-consistency statement:
  * INTER_LINEAR with half-pixel centres and replicated borders  <->  torch's CPU upsample_bilinear2d(align_corners=False),
    which implements src = (dst + 0.5) * scale - 0.5 clamped at the borders in its own (non-separable, double-free)
    arithmetic: values agree to float32 rounding (<= 2e-6 relative to the value range), binarised masks agree except where
    the interpolated value is within that rounding of the threshold;
  * the COCO RLE format  <->  a decoder written here straight from the format's definition (column-major runs starting
    with a run of zeros; counts string = 5-bit groups, bit 5 = continuation, bit 4 of the last group = sign, every count
    from the third on stored as the difference to the count two places back): decode(encode(mask)) == mask on random and
    degenerate masks, i.e. the encoder emits valid, loss-free COCO RLE.
CPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------- A9: mask resize -------------------------------------------
def _torch_mask_values(mask, box, M):
    """What result_utils.py:185-202 computes, through torch's bilinear: zero-pad to (M+2)^2, resize to (h, w)."""
    w, h = max(int(box[2] - box[0] + 1), 1), max(int(box[3] - box[1] + 1), 1)
    padded = np.zeros((M + 2, M + 2), np.float32)
    padded[1:-1, 1:-1] = mask
    t = torch.from_numpy(padded)[None, None]
    return F.interpolate(t, size=(h, w), mode="bilinear", align_corners=False)[0, 0].numpy()


@pytest.mark.parametrize("M", [14, 28])
def test_mask_resize_vs_torch_bilinear(oracle, M):
    rs = np.random.RandomState(900 + M)
    n_diff = n_tot = 0
    for k in range(40):
        mask = (1.0 / (1.0 + np.exp(-rs.randn(M, M) * 1.5))).astype(np.float32)
        cx, cy = rs.uniform(40, 760), rs.uniform(40, 460)
        bw, bh = np.exp(rs.uniform(np.log(3), np.log(400))), np.exp(rs.uniform(np.log(3), np.log(300)))   # up- and down-scaling
        ref_box = np.array([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], np.float32)
        box, crop = oracle.mask_resize_binarize(mask, ref_box, 0.5)
        vals = _torch_mask_values(mask, box, M)
        assert crop.shape == vals.shape
        want = (vals > 0.5).astype(np.uint8)
        diff = crop != want
        # every disagreement sits on the threshold to float32 rounding of two different evaluation orders
        assert np.all(np.abs(vals[diff] - 0.5) <= 2e-6), (k, np.abs(vals[diff] - 0.5).max())
        n_diff += int(diff.sum()); n_tot += diff.size
    assert n_tot > 100000 and n_diff <= n_tot * 1e-4, (n_diff, n_tot)


# ---------------------------------------------------------------- f-3: input preparation ------------------------------------
@pytest.mark.parametrize("h,w", [(480, 640), (375, 500), (1200, 1600), (333, 1000)])
def test_prep_image_vs_torch_bilinear(oracle, h, w):
    rs = np.random.RandomState(h + w)
    im = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    means = (122.7717, 115.9465, 102.9801)
    blob, scales = oracle.prep_images([im], means, 800, 1333, 32)
    s = scales[0]
    oh, ow = max(int(np.round(h * s)), 1), max(int(np.round(w * s)), 1)
    # blob.py:72-73 then cv2.resize(fx=fy=im_scale): coordinates use 1/scale -> recompute_scale_factor=False
    x = torch.from_numpy(im.astype(np.float32) - np.asarray(means, np.float32)).permute(2, 0, 1)[None]
    y = F.interpolate(x, scale_factor=(s, s), mode="bilinear", align_corners=False, recompute_scale_factor=False)[0].numpy()
    # torch sizes the output with floor(in * scale), OpenCV with round: compare the common region
    hh, ww = min(oh, y.shape[1]), min(ow, y.shape[2])
    assert hh >= oh - 1 and ww >= ow - 1
    got = blob[0, :, :hh, :ww]
    # both sides form the source coordinate in float32 (OpenCV: `fx = (float)((dx + 0.5) * scale_x - 0.5)`; torch: all-float32
    # arithmetic on 1/scale): near column 1000 one ulp of the coordinate is 6e-5, times a neighbour difference of up to 255
    err = np.abs(got - y[:, :hh, :ww])
    assert err.max() <= 255 * 2.5e-4 and err.mean() <= 2e-3, (err.max(), err.mean())
    assert not blob[0, :, oh:, :].any() and not blob[0, :, :, ow:].any()     # zero padding (blob.py:45-57)


# ---------------------------------------------------------------- f-4: COCO RLE --------------------------------------------
def _rle_string_decode(s):
    cnts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1; k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    return cnts


def _rle_runs_decode(runs, h, w):
    flat = np.zeros(h * w, np.uint8)
    pos, v = 0, 0
    for r in runs:
        flat[pos:pos + r] = v
        pos += r; v ^= 1
    assert pos == h * w
    return flat.reshape(w, h).T            # column-major


@pytest.mark.parametrize("case", ["random", "blobs", "empty", "full", "first_set", "single_column", "big_runs"])
def test_rle_round_trip_through_independent_decoder(oracle, case):
    rs = np.random.RandomState(sum(map(ord, case)))
    if case == "random":
        masks = [(rs.rand(37, 53) > p).astype(np.uint8) for p in (0.1, 0.5, 0.9)]
    elif case == "blobs":
        masks = []
        for _ in range(3):
            m = np.zeros((120, 90), np.uint8)
            for _ in range(4):
                y, x = rs.randint(0, 100), rs.randint(0, 70)
                m[y:y + rs.randint(1, 40), x:x + rs.randint(1, 40)] = 1
            masks.append(m)
    elif case == "empty":
        masks = [np.zeros((9, 7), np.uint8)]
    elif case == "full":
        masks = [np.ones((9, 7), np.uint8)]
    elif case == "first_set":
        m = np.zeros((5, 4), np.uint8); m[0, 0] = 1; masks = [m]
    elif case == "single_column":
        masks = [(rs.rand(200, 1) > 0.5).astype(np.uint8), (rs.rand(1, 200) > 0.5).astype(np.uint8)]
    else:   # runs that need 3-4 five-bit groups, and negative differences
        m = np.zeros((500, 833), np.uint8); m[100:400, 200:600] = 1; m[250, 300:500] = 0; masks = [m]
    for m in masks:
        runs, s = oracle.rle_encode(m)
        h, w = m.shape
        assert int(runs.sum()) == h * w
        assert all(48 <= ord(c) < 48 + 64 for c in s)
        assert _rle_string_decode(s) == [int(r) for r in runs]
        assert np.array_equal(_rle_runs_decode([int(r) for r in runs], h, w), m)
