"""Opt-in pinning of the three "parity unpinned" legs (A9 mask resize, f-3 input preparation, f-4 RLE) against goldens produced by
the reference's own third-party dependencies -- tests/golden/make_thirdparty_golden.py, which needs cv2 / pycocotools and
therefore cannot run in this image.  Every test is SKIPPED while its .npz is absent; once someone with those libraries commits
the files, the CPU tests pin the oracle and the GPU tests pin the HIP kernels, and the rows can leave "partial"."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    path = os.path.join(GOLD, name)
    if not os.path.exists(path):
        pytest.skip("%s not committed (run tests/golden/make_thirdparty_golden.py where cv2 / pycocotools exist)" % name)
    return np.load(path)


def _n(g, prefix):
    k = 0
    while "%s%d" % (prefix, k) in g:
        k += 1
    return k


# ---- CPU: the oracle against the third-party outputs -------------------------------------------------------------------------
def test_oracle_mask_resize_vs_cv2(oracle):
    g = _load("mask_resize_cv2.npz")
    n_diff = n_tot = 0
    for k in range(_n(g, "mask")):
        box, crop = oracle.mask_resize_binarize(g["mask%d" % k], g["ref_box%d" % k], 0.5)
        assert np.array_equal(box, g["box%d" % k]), k                        # expand_boxes + int32 truncation
        want, vals = g["binary%d" % k], g["resized%d" % k]
        assert crop.shape == want.shape, k
        diff = crop != want
        # OpenCV evaluates INTER_LINEAR in its own order (fixed-point for 8-bit, float32 vertical-then-horizontal here):
        # a disagreement is legitimate only where the interpolated value sits on the threshold to float32 rounding
        assert np.all(np.abs(vals[diff] - 0.5) <= 4e-6), (k, np.abs(vals[diff] - 0.5).max())
        n_diff += int(diff.sum()); n_tot += diff.size
    assert n_diff <= n_tot * 1e-4, (n_diff, n_tot)


def test_oracle_prep_vs_cv2(oracle):
    g = _load("prep_cv2.npz")
    means = (122.7717, 115.9465, 102.9801)
    for k in range(_n(g, "image")):
        blob, scales = oracle.prep_images([g["image%d" % k]], means, 800, 1333, 32)
        assert scales[0] == float(g["scale%d" % k]), k
        y = np.transpose(g["resized%d" % k], (2, 0, 1))
        got = blob[0, :, :y.shape[1], :y.shape[2]]
        err = np.abs(got - y)
        assert err.max() <= 255 * 2.5e-4, (k, err.max())                     # one float32 ulp of the source coordinate
        assert not blob[0, :, y.shape[1]:, :].any() and not blob[0, :, :, y.shape[2]:].any()


def test_oracle_rle_vs_pycocotools(oracle):
    g = _load("rle_pycocotools.npz")
    for k in range(_n(g, "mask")):
        runs, s = oracle.rle_encode(g["mask%d" % k])
        assert s.encode("ascii") == g["counts%d" % k].tobytes(), k           # the very bytes pycocotools emits


# ---- GPU: the HIP kernels against the same files --------------------------------------------------------------------------------
@pytest.mark.gpu
def test_hip_mask_paste_vs_cv2(oracle):
    import torch
    from detectorch_amd import hip
    g = _load("mask_resize_cv2.npz")
    for M in (14, 28):
        ks = [k for k in range(_n(g, "mask")) if g["mask%d" % k].shape[0] == M]
        D = len(ks)
        masks = torch.zeros((D, 2, M, M), device="cuda")
        dets = torch.zeros((1, D, 6), device="cuda")
        for i, k in enumerate(ks):
            masks[i, 1] = torch.from_numpy(g["mask%d" % k]).cuda()
            dets[0, i, :4] = torch.from_numpy(g["ref_box%d" % k]).cuda()
            dets[0, i, 4] = 0.9; dets[0, i, 5] = 1
        cnt = torch.tensor([D], dtype=torch.int32, device="cuda")
        imsz = torch.tensor([[2000.0, 2000.0]], device="cuda")
        cap = 64 << 20
        crops = torch.zeros((1, cap), dtype=torch.uint8, device="cuda")
        boxes = torch.zeros((1, D, 4), dtype=torch.int32, device="cuda"); rects = torch.zeros_like(boxes)
        offs = torch.zeros((1, D), dtype=torch.int64, device="cuda"); nbytes = torch.zeros((1,), dtype=torch.int64, device="cuda")
        hip.check(hip.lib().dtc_mask_paste(masks.data_ptr(), None, 2, M, dets.data_ptr(), cnt.data_ptr(), imsz.data_ptr(), 1, D, 0.5, 1,
                                           crops.data_ptr(), cap, boxes.data_ptr(), rects.data_ptr(), offs.data_ptr(), nbytes.data_ptr(),
                                           hip.stream_ptr()), "mask_paste")
        torch.cuda.synchronize()
        cr, bx, rc, of = crops.cpu().numpy()[0], boxes.cpu().numpy()[0], rects.cpu().numpy()[0], offs.cpu().numpy()[0]
        n_diff = n_tot = 0
        for i, k in enumerate(ks):
            box, want, vals = g["box%d" % k], g["binary%d" % k], g["resized%d" % k]
            assert np.array_equal(bx[i], box), k
            x0, y0, x1, y1 = rc[i]                                            # paste rectangle inside the (2000, 2000) frame
            got = cr[of[i]:of[i] + (x1 - x0) * (y1 - y0)].reshape(y1 - y0, x1 - x0)
            sub = (slice(y0 - box[1], y1 - box[1]), slice(x0 - box[0], x1 - box[0]))
            diff = got != want[sub]
            assert np.all(np.abs(vals[sub][diff] - 0.5) <= 4e-6), k
            n_diff += int(diff.sum()); n_tot += diff.size
        assert n_diff <= n_tot * 1e-4


@pytest.mark.gpu
def test_hip_prep_vs_cv2(oracle):
    import torch
    from detectorch_amd.utils import blob as blob_utils
    g = _load("prep_cv2.npz")
    for k in range(_n(g, "image")):
        im = g["image%d" % k]
        out, scales = blob_utils.images_to_blob([im], target_size=800, max_size=1333, fpn_on=True)
        y = np.transpose(g["resized%d" % k], (2, 0, 1))
        got = out[0, :, :y.shape[1], :y.shape[2]].cpu().numpy()
        assert scales[0] == float(g["scale%d" % k])
        assert np.abs(got - y).max() <= 255 * 2.5e-4, k


@pytest.mark.gpu
def test_hip_rle_vs_pycocotools(oracle):
    import torch
    from detectorch_amd import hip
    g = _load("rle_pycocotools.npz")
    for k in range(_n(g, "mask")):
        m = g["mask%d" % k]
        h, w = m.shape
        crops = torch.from_numpy(np.ascontiguousarray(m)).cuda().reshape(1, -1)
        rects = torch.tensor([[[0, 0, w, h]]], dtype=torch.int32, device="cuda")
        offs = torch.zeros((1, 1), dtype=torch.int64, device="cuda")
        cnt = torch.ones((1,), dtype=torch.int32, device="cuda")
        imsz = torch.tensor([[float(h), float(w)]], device="cuda")
        RS, SS = max(h * w + 2, 16), max(2 * h * w + 16, 64)
        runs = torch.zeros((1, 1, RS), dtype=torch.int32, device="cuda"); nr = torch.zeros((1, 1), dtype=torch.int32, device="cuda")
        st = torch.zeros((1, 1, SS), dtype=torch.uint8, device="cuda"); sl = torch.zeros((1, 1), dtype=torch.int32, device="cuda")
        hip.check(hip.lib().dtc_mask_rle(crops.data_ptr(), crops.numel(), rects.data_ptr(), offs.data_ptr(), cnt.data_ptr(), imsz.data_ptr(),
                                         1, 1, runs.data_ptr(), RS, nr.data_ptr(), st.data_ptr(), SS, sl.data_ptr(), hip.stream_ptr()),
                  "mask_rle")
        torch.cuda.synchronize()
        n = int(sl[0, 0])
        assert n > 0 and st[0, 0, :n].cpu().numpy().tobytes() == g["counts%d" % k].tobytes(), k
