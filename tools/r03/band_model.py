"""Offline model of the band-sweep RoIAlign kernel's control logic on the bench RoI distribution (tools/r02b/gen_rois.py
writes /tmp/rois_<i>.npy): band items, batches, staged pieces, eligibility for a given LDS row capacity."""
import sys
import numpy as np

F = np.float32


def axis(start, binsz, p, i, grid, extent):
    v = F(start) + F(p) * F(binsz)
    v = F(v + F(F(F(i) + F(.5)) * F(binsz)) / F(grid))
    if v <= 0: v = F(0)
    lo = int(v)
    if lo >= extent - 1: lo = hi = extent - 1
    else: hi = lo + 1
    return lo, hi


def windows(rois, shapes, scales):
    out = []
    for r in rois:
        l = int(r[5]); H, W = shapes[l]; s = F(scales[l])
        sw, sh = F(r[1]) * s, F(r[2]) * s
        rw = max(F(F(r[3]) * s - sw), F(1)); rh = max(F(F(r[4]) * s - sh), F(1))
        bh, bw = F(rh / F(7)), F(rw / F(7))
        y0 = axis(sh, bh, 0, 0, 2, H)[0]; y1 = axis(sh, bh, 6, 1, 2, H)[1]
        x0 = axis(sw, bw, 0, 0, 2, W)[0]; x1 = axis(sw, bw, 6, 1, 2, W)[1]
        out.append((l, x0, x1, y0, y1))
    return np.array(out)


def main(n_img, band_log2=5, ring=64, K=20, rows_cap=54):
    shapes = [(200, 336), (100, 168), (50, 84), (25, 42)]
    scales = [1 / 4., 1 / 8., 1 / 16., 1 / 32.]
    tot = dict(rois=0, items=0, batches=0, inel=0, pieces=0, staged_px=0, bat_sz=[], units=[], span=[])
    for i in range(n_img):
        rois = np.load("/tmp/rois_%d.npy" % i)
        w = windows(rois, shapes, scales)
        fs = w[:, 0] + 2
        yc = ((rois[:, 2] + rois[:, 4]) * 0.5).astype(np.int64); xc = ((rois[:, 1] + rois[:, 3]) * 0.5).astype(np.int64)
        band = (yc >> fs) >> band_log2
        order = np.lexsort((np.arange(len(rois)), xc >> fs, band, w[:, 0]))
        w, band = w[order], band[order]
        key = w[:, 0] * 64 + band
        starts = np.flatnonzero(np.r_[True, key[1:] != key[:-1]])
        ends = np.r_[starts[1:], len(key)]
        tot["rois"] += len(key)
        for s, e in zip(starts, ends):
            m = w[s:e]
            rbase = m[:, 3].min(); span = m[:, 4].max() - rbase + 1
            tot["items"] += 1; tot["span"].append(span)
            elig = (m[:, 4] - rbase < rows_cap) & (((m[:, 2] | 3) - (m[:, 1] & ~3) + 1) <= ring)
            tot["inel"] += int((~elig).sum())
            rows = min(span, rows_cap)
            j = 0; res_a = res_b = 0
            while j < len(m):
                if not elig[j]: j += 1; continue
                xa, xb = m[j, 1] & ~3, m[j, 2] | 3; n = 1
                while j + n < len(m) and n < K and elig[j + n]:
                    a, b = min(xa, m[j + n, 1] & ~3), max(xb, m[j + n, 2] | 3)
                    if b - a + 1 > ring: break
                    xa, xb = a, b; n += 1
                ka, kb = max(xa, res_a), min(xb + 1, res_b)
                newg = ((xb + 1 - xa) - max(0, kb - ka)) // 4
                units = 2 * rows * ((newg + 15) // 16) if newg else 0
                tot["units"].append(units); tot["pieces"] += newg * rows * 8; tot["staged_px"] += newg * 4 * rows
                tot["batches"] += 1; tot["bat_sz"].append(n)
                res_a, res_b = xa, xb + 1
                j += n
    bs = np.array(tot["bat_sz"]); un = np.array(tot["units"]); sp = np.array(tot["span"])
    print("band_log2 %d ring %d K %d rows_cap %d: rois %d items %d (%.1f/img) batches %d (mean size %.1f, lanes used %.0f%%) ineligible %d"
          % (band_log2, ring, K, rows_cap, tot["rois"], tot["items"], tot["items"] / n_img, tot["batches"], bs.mean(), 100 * bs.mean() * 49 / 1024, tot["inel"]))
    print("  row span of an item: mean %.1f max %d ; wave-units per batch (16 waves x U): mean %.1f max %d -> U mean %.1f max %.1f"
          % (sp.mean(), sp.max(), un.mean(), un.max(), un.mean() / 16, un.max() / 16))
    print("  staged pixels per (RoI, channel): %.1f  -> line fills per (RoI, channel) ~ %.2f ; feature bytes per launch of 8 images x 256 ch: %.2f GB"
          % (tot["staged_px"] / tot["rois"], tot["staged_px"] / tot["rois"] / 32, tot["staged_px"] / n_img * 8 * 256 * 4 / 1e9))


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    for bl, ring, K, cap in ((5, 64, 20, 54), (5, 64, 20, 60), (5, 64, 16, 56), (4, 64, 20, 40), (5, 48, 20, 72), (6, 64, 20, 90)):
        main(n, bl, ring, K, cap)
