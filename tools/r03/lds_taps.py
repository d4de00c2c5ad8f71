"""Offline: LDS cycles of the band kernel's tap gather (ds_read_b128, CDNA4 lane groups of 16, 16 slots of 16 B) on the bench RoI
distribution, for candidate ring layouts: slot(row, col, quad) -> average cycles per wave-instruction (4 = conflict-free)."""
import sys
import numpy as np
sys.path.insert(0, "tools/r03")
from band_model import axis, F

GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def roi_taps(r, shapes, scales):
    l = int(r[5]); H, W = shapes[l]; s = F(scales[l])
    sw, sh = F(r[1]) * s, F(r[2]) * s
    rw = max(F(F(r[3]) * s - sw), F(1)); rh = max(F(F(r[4]) * s - sh), F(1))
    bh, bw = F(rh / F(7)), F(rw / F(7))
    ys = [[axis(sh, bh, p, i, 2, H) for i in range(2)] for p in range(7)]
    xs = [[axis(sw, bw, p, i, 2, W) for i in range(2)] for p in range(7)]
    return ys, xs


def main():
    shapes = [(200, 336), (100, 168), (50, 84), (25, 42)]
    scales = [1 / 4., 1 / 8., 1 / 16., 1 / 32.]
    rois = np.load("/tmp/rois_0.npy")
    lv = rois[:, 5].astype(int)
    fs = lv + 2
    yc = ((rois[:, 2] + rois[:, 4]) * 0.5).astype(np.int64); xc = ((rois[:, 1] + rois[:, 3]) * 0.5).astype(np.int64)
    band = (yc >> fs) >> 5
    order = np.lexsort((np.arange(len(rois)), xc >> fs, band, lv))
    rois = rois[order]
    taps = [roi_taps(r, shapes, scales) for r in rois[:400]]            # the first P2 bands
    layouts = {
        "current: slot = phys(col) + 8q (row pitch 2304)": lambda row, col, q: (col & 63) + ((col & 63) >> 3) + 8 * q,
        "row skew 1": lambda row, col, q: (col & 63) + ((col & 63) >> 3) + 8 * q + row,
        "row skew 3": lambda row, col, q: (col & 63) + ((col & 63) >> 3) + 8 * q + 3 * row,
        "row skew 5": lambda row, col, q: (col & 63) + ((col & 63) >> 3) + 8 * q + 5 * row,
        "row skew 7": lambda row, col, q: (col & 63) + ((col & 63) >> 3) + 8 * q + 7 * row,
        "no pad, skew 1": lambda row, col, q: (col & 63) + 8 * q + row,
        "no pad, skew 5": lambda row, col, q: (col & 63) + 8 * q + 5 * row,
        "xor skew": lambda row, col, q: ((col & 63) + ((col & 63) >> 3) + 8 * q) ^ (row & 15),
    }
    for bins_per in (49,):
        for name, f in layouts.items():
            tot, n = 0, 0
            for b0 in range(0, 400, 20):                     # batches of 20 consecutive RoIs
                items = [(k, bn) for k in range(b0, min(b0 + 20, len(taps))) for bn in range(49)]
                for w0 in range(0, len(items), 64):
                    wave = items[w0:w0 + 64]
                    for iy in range(2):
                        for ix in range(2):
                            for tap in range(4):
                                cyc = 0
                                for g in GROUPS:
                                    occ = {}
                                    for ln in g:
                                        if ln >= len(wave): continue
                                        k, bn = wave[ln]
                                        ph, pw = divmod(bn, 7)
                                        ylo, yhi = taps[k][0][ph][iy]; xlo, xhi = taps[k][1][pw][ix]
                                        row = (ylo, ylo, yhi, yhi)[tap]; col = (xlo, xhi, xlo, xhi)[tap]
                                        s = f(row, col, 0) & 15
                                        occ.setdefault(s, set()).add((row, col))
                                    cyc += max([len(v) for v in occ.values()], default=1)
                                tot += cyc; n += 1
            print("%-52s %.2f LDS cycles per ds_read_b128 (ideal 4)" % (name, tot / n))


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def cluster_layouts():
    """The cluster-stationary kernel's image: px = row * tw + col inside the union window of 5 consecutive RoIs, slot = px + px / 8
    (one pad slot per 8 pixels); alternatives: row pitch padded so that consecutive rows advance the slot by an odd amount."""
    shapes = [(200, 336), (100, 168), (50, 84), (25, 42)]
    scales = [1 / 4., 1 / 8., 1 / 16., 1 / 32.]
    rois = np.load("/tmp/rois_0.npy")                      # already in the 16-row-band visiting order
    taps = [roi_taps(r, shapes, scales) for r in rois[:400]]
    res = {}
    for name in ("kernel: phys(row * tw + col)", "row pitch = tw + tw/8 rounded up to 16k + 5 slots", "row pitch 16k + 7", "row pitch 16k + 3"):
        tot, n = 0, 0
        for b0 in range(0, 400, 5):
            grp = list(range(b0, b0 + 5))
            x0 = min(taps[k][1][0][0][0] for k in grp) & ~3
            x1 = max(taps[k][1][6][1][1] for k in grp)
            y0 = min(taps[k][0][0][0][0] for k in grp)
            tw = 4 * ((x1 >> 2) - (x0 >> 2) + 1)
            if name.startswith("kernel"):
                f = lambda row, col: (lambda px: px + (px >> 3))((row - y0) * tw + (col - x0))
            else:
                skew = int(name.split("+")[-1].split()[0]) if "16k" in name else 5
                base = tw + (tw >> 3)
                pitch = ((base - skew + 15) // 16) * 16 + skew
                f = lambda row, col: (row - y0) * pitch + (col - x0) + ((col - x0) >> 3)
            items = [(k, bn) for k in grp for bn in range(49)]
            for w0 in range(0, len(items), 64):
                wave = items[w0:w0 + 64]
                for iy in range(2):
                    for ix in range(2):
                        for tap in range(4):
                            cyc = 0
                            for g in GROUPS:
                                occ = {}
                                for ln in g:
                                    if ln >= len(wave): continue
                                    k, bn = wave[ln]
                                    ph, pw = divmod(bn, 7)
                                    ylo, yhi = taps[k][0][ph][iy]; xlo, xhi = taps[k][1][pw][ix]
                                    row = (ylo, ylo, yhi, yhi)[tap]; col = (xlo, xhi, xlo, xhi)[tap]
                                    occ.setdefault(f(row, col) & 15, set()).add((row, col))
                                cyc += max([len(v) for v in occ.values()], default=1)
                            tot += cyc; n += 1
        print("cluster image, %-55s %.2f LDS cycles per ds_read_b128" % (name, tot / n))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "cluster":
    cluster_layouts()
