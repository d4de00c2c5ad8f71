"""Pin-when-available goldens for the three legs whose arithmetic lives in third-party code that is NOT in this image
(SURVEY 8c): OpenCV's cv2.resize (mask paste: lib/utils/result_utils.py:197-203; input preparation: lib/utils/blob.py:75-87)
and pycocotools' RLE (result_utils.py:217-220).  Run this ON A HOST THAT HAS cv2 AND/OR pycocotools:

    python tests/golden/make_thirdparty_golden.py            # writes tests/golden/{mask_resize_cv2,prep_cv2,rle_pycocotools}.npz

It calls the very functions the reference calls, on seeded inputs (make_inputs below, shared with tests/test_thirdparty_golden.py),
and stores inputs + outputs + the library versions.  tests/test_thirdparty_golden.py then compares the oracle (CPU) and the HIP
kernels (GPU) with these files and is skipped while they are absent -- the rows A9 / f-3 / f-4 stay "parity unpinned" until
someone commits them.  Nothing here imports the product or the oracle."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def mask_inputs():
    """40 (mask, reference box) pairs per mask size: sigmoid(N(0, 1.5)) masks, boxes from 3 to 400 pixels, some over the border."""
    out = []
    for M in (14, 28):
        rs = np.random.RandomState(900 + M)
        for _ in range(40):
            mask = (1.0 / (1.0 + np.exp(-rs.randn(M, M) * 1.5))).astype(np.float32)
            cx, cy = rs.uniform(-20, 860), rs.uniform(-20, 520)
            bw, bh = np.exp(rs.uniform(np.log(3), np.log(400))), np.exp(rs.uniform(np.log(3), np.log(300)))
            out.append((M, mask, np.array([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], np.float32)))
    return out


def prep_inputs():
    rs = np.random.RandomState(77)
    return [rs.randint(0, 256, (h, w, 3)).astype(np.uint8) for (h, w) in ((480, 640), (375, 500), (600, 1000), (333, 1000))]


def rle_inputs():
    rs = np.random.RandomState(78)
    masks = [(rs.rand(37, 53) > p).astype(np.uint8) for p in (0.1, 0.5, 0.9)]
    for _ in range(3):
        m = np.zeros((120, 90), np.uint8)
        for _ in range(4):
            y, x = rs.randint(0, 100), rs.randint(0, 70)
            m[y:y + rs.randint(1, 40), x:x + rs.randint(1, 40)] = 1
        masks.append(m)
    masks += [np.zeros((9, 7), np.uint8), np.ones((9, 7), np.uint8)]
    m = np.zeros((500, 833), np.uint8); m[100:400, 200:600] = 1; m[250, 300:500] = 0
    masks.append(m)
    return masks


def expand_and_truncate(ref_box, M):
    """expand_boxes(ref_boxes, (M + 2) / M).astype(np.int32): lib/utils/boxes.py:245-261, result_utils.py:182-184"""
    scale = np.float32((M + 2.0) / M)
    b = ref_box.astype(np.float32)[None]
    w_half, h_half = (b[:, 2] - b[:, 0]) * np.float32(.5), (b[:, 3] - b[:, 1]) * np.float32(.5)
    x_c, y_c = (b[:, 2] + b[:, 0]) * np.float32(.5), (b[:, 3] + b[:, 1]) * np.float32(.5)
    w_half = w_half * scale; h_half = h_half * scale
    e = np.zeros(b.shape, np.float32)
    e[:, 0] = x_c - w_half; e[:, 2] = x_c + w_half; e[:, 1] = y_c - h_half; e[:, 3] = y_c + h_half
    return e.astype(np.int32)[0]


def main():
    done = []
    try:
        import cv2
        # ---- A9: result_utils.py:185-203 ---------------------------------------------------------------------------------
        rec = {"cv2_version": np.array(cv2.__version__)}
        for k, (M, mask, ref_box) in enumerate(mask_inputs()):
            box = expand_and_truncate(ref_box, M)
            padded = np.zeros((M + 2, M + 2), np.float32)
            padded[1:-1, 1:-1] = mask
            w, h = max(box[2] - box[0] + 1, 1), max(box[3] - box[1] + 1, 1)
            resized = cv2.resize(padded, (int(w), int(h)))                     # INTER_LINEAR, :202
            rec["mask%d" % k] = mask; rec["ref_box%d" % k] = ref_box; rec["box%d" % k] = box
            rec["resized%d" % k] = resized.astype(np.float32)
            rec["binary%d" % k] = np.array(resized > 0.5, np.uint8)            # :203
        np.savez_compressed(os.path.join(HERE, "mask_resize_cv2.npz"), **rec)
        done.append("mask_resize_cv2.npz")
        # ---- f-3: blob.py:62-87 (prep_im_for_blob, target 800 / max 1333) --------------------------------------------------
        rec = {"cv2_version": np.array(cv2.__version__)}
        for k, im in enumerate(prep_inputs()):
            x = im.astype(np.float32, copy=False)
            x -= [122.7717, 115.9465, 102.9801]               # blob.py:65-66, verbatim (in-place on the float32 image)
            short, long_ = min(x.shape[:2]), max(x.shape[:2])
            scale = float(800) / float(short)
            if np.round(scale * long_) > 1333:
                scale = float(1333) / float(long_)
            y = cv2.resize(x, None, None, fx=scale, fy=scale, interpolation=cv2.INTER_LINEAR)
            rec["image%d" % k] = im; rec["scale%d" % k] = np.float64(scale); rec["resized%d" % k] = y.astype(np.float32)
        np.savez_compressed(os.path.join(HERE, "prep_cv2.npz"), **rec)
        done.append("prep_cv2.npz")
    except ImportError:
        print("cv2 not importable here: mask_resize_cv2.npz / prep_cv2.npz not written")
    try:
        import pycocotools.mask as mask_util
        rec = {}
        for k, m in enumerate(rle_inputs()):
            rle = mask_util.encode(np.array(m[:, :, np.newaxis], order='F'))[0]      # result_utils.py:217-220
            counts = rle["counts"]
            rec["mask%d" % k] = m
            rec["counts%d" % k] = np.frombuffer(counts if isinstance(counts, bytes) else counts.encode("ascii"), np.uint8)
            rec["size%d" % k] = np.asarray(rle["size"], np.int64)
        np.savez_compressed(os.path.join(HERE, "rle_pycocotools.npz"), **rec)
        done.append("rle_pycocotools.npz")
    except ImportError:
        print("pycocotools not importable here: rle_pycocotools.npz not written")
    print("written:", done if done else "nothing (neither library is importable)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
