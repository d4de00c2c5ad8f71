"""CPU emulation of the band-sweep draft's control logic (tools/r03_draft/roi_align_band.hip): run table, batch selection
(prefix min / max over 64 candidates), ring residency and tap addressing -- on the bench's RoI distribution
(python tools/r02b/gen_rois.py 4 first).  Asserts that every tap of every RoI reads a ring slot that holds its column and a row
inside the staged range; prints the staged pixels per (RoI, channel) and the batch statistics."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools", "r02"))
from analyze_clusters import axis, SCALES, SHAPES

RING, BAND_LOG2, P = 64, 5, 7


def geom(d):
    _, x1, y1, x2, y2, lvl = d[:6]
    lvl = int(lvl)
    s = np.float32(SCALES[lvl]); H, W = SHAPES[lvl]
    sw, sh = np.float32(x1) * s, np.float32(y1) * s
    rw, rh = max(np.float32(x2) * s - sw, np.float32(1)), max(np.float32(y2) * s - sh, np.float32(1))
    return lvl, H, W, sw, sh, rw / np.float32(P), rh / np.float32(P)


def main(n_img):
    staged_px = rois = nb = 0
    sizes = []
    for img in range(n_img):
        d = np.load("/tmp/rois_%d.npy" % img)
        # visiting order with 32-row bands: (level, band of the centre row, x centre) -- fpn.hip's key
        lvl = d[:, 5].astype(np.int64); fs = lvl + 2
        yc = ((d[:, 2] + d[:, 4]) * 0.5).astype(np.int64); xc = ((d[:, 1] + d[:, 3]) * 0.5).astype(np.int64)
        band = np.minimum((yc >> fs) >> BAND_LOG2, 63); xf = np.minimum(xc >> fs, 4095)
        order = np.lexsort((np.arange(len(d)), xf, band, lvl))
        d, lvl, band = d[order], lvl[order], band[order]
        key = lvl * 64 + band
        starts = [0] + [i for i in range(1, len(d)) if key[i] != key[i - 1]] + [len(d)]
        for a, b in zip(starts[:-1], starts[1:]):                      # one band instance
            G = [geom(r) for r in d[a:b]]
            _, H, W = G[0][:3]
            wins = []
            for (_, H, W, sw, sh, bw, bh) in G:
                wins.append((axis(sw, bw, 0, 0, 2, W)[0], axis(sw, bw, P - 1, 1, 2, W)[1], axis(sh, bh, 0, 0, 2, H)[0], axis(sh, bh, P - 1, 1, 2, H)[1]))
            rbase, rmax = min(w[2] for w in wins), max(w[3] for w in wins)
            rows = rmax - rbase + 1
            slot_col = [-1] * RING                                      # which column each ring slot holds
            res_a = res_b = 0
            i0 = 0
            while i0 < len(wins):
                mn, mx, n = 1 << 30, -1, 0
                for k in range(min(64, len(wins) - i0)):               # prefix min / max, leading run that fits
                    x0, x1 = wins[i0 + k][0] & ~3, wins[i0 + k][1] | 3
                    mn2, mx2 = min(mn, x0), max(mx, x1)
                    if mx2 - mn2 + 1 > RING: break
                    mn, mx, n = mn2, mx2, n + 1
                assert n >= 1, "a window wider than the ring"
                xa, xb = mn, min(mx, (W - 1) | 3)
                keep_a, keep_b = max(xa, res_a), min(xb + 1, res_b)
                for col in range(xa, xb + 1):
                    if keep_b > keep_a and keep_a <= col < keep_b:
                        assert slot_col[col & (RING - 1)] == col       # claimed resident: must really be there
                        continue
                    slot_col[col & (RING - 1)] = col
                    staged_px += rows
                res_a, res_b = xa, xb + 1
                for k in range(n):                                      # every tap of the batch's RoIs
                    _, H, W, sw, sh, bw, bh = G[i0 + k]
                    for p in range(P):
                        for i in range(2):
                            for v in axis(sw, bw, p, i, 2, W):
                                assert slot_col[v & (RING - 1)] == v, (img, a, i0 + k, v)
                            for v in axis(sh, bh, p, i, 2, H):
                                assert rbase <= v <= rmax
                sizes.append(n); nb += 1
                i0 += n
            rois += len(wins)
    print("%d RoIs, %d batches (mean %.1f RoIs), staged pixels per (RoI, channel): %.0f  (cluster kernel: ~316)"
          % (rois, nb, np.mean(sizes), staged_px / rois))
    print("every tap found its column in the ring and its row in the staged range")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 2)
