"""Development aid (offline, CPU): L2->L1 line fills per (RoI, channel) of the FPN box-head RoIAlign launch under different
staging structures, on the bench's RoI distribution (tools/r02b/gen_rois.py writes /tmp/rois_<i>.npy).  fp32 NCHW maps, 128-byte
lines.  Models:
  compulsory      every line any window touches, once per (image, channel)                       -- the floor
  roi             one RoI's window per workgroup (round 1: roi_align_fwd_lds)
  cluster K       K consecutive RoIs of the visiting order merged greedily (roi_align_fwd_tile, merge rule 250 %)
  band H          a workgroup owns a band of H feature rows of one level x a channel block and sweeps it in x with a sliding
                  LDS window: it stages the rows its RoIs' windows cover (RoIs are assigned to a band by their CENTRE row, as
                  in the visiting order: the windows stick out above and below the band) once per channel
Usage: python tools/r02b/gen_rois.py 4 && python tools/r02b/fill_models.py 4"""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 3)[0] + "/tools/r02")
from analyze_clusters import window, SHAPES


def lines_of(row, x0, x1, W):
    b0, b1 = (row * W + x0) * 4, (row * W + x1) * 4 + 3
    return range(b0 >> 7, (b1 >> 7) + 1)


def main(n_img):
    tot = dict(rois=0, compulsory=0, roi=0)
    clus = {5: 0, 10: 0, 20: 0}
    bands = {8: 0, 16: 0, 32: 0, 64: 0}
    band_lds = {h: 0 for h in bands}
    for i in range(n_img):
        d = np.load("/tmp/rois_%d.npy" % i)
        wins = [window(r) for r in d]
        tot["rois"] += len(wins)
        touched = [set() for _ in SHAPES]
        for w in wins:
            _, lvl, x0, x1, y0, y1 = w
            W = SHAPES[lvl][1]
            for row in range(y0, y1 + 1):
                ls = lines_of(row, x0, x1, W)
                touched[lvl].update(ls)
                tot["roi"] += len(ls)
        tot["compulsory"] += sum(len(t) for t in touched)
        # clusters of K consecutive RoIs (visiting order), union bounding patch, 4-pixel aligned columns, merge rule 250 %
        for K in clus:
            for g0 in range(0, len(wins), K):
                grp = wins[g0:g0 + K]
                k = 0
                while k < len(grp):
                    a = grp[k]; x0, x1, y0, y1 = a[2], a[3], a[4], a[5]; cnt = 1
                    spx = (y1 - y0 + 1) * (x1 - x0 + 1)
                    while k + cnt < len(grp):
                        n = grp[k + cnt]
                        if n[1] != a[1]: break
                        ux0, ux1, uy0, uy1 = min(x0, n[2]), max(x1, n[3]), min(y0, n[4]), max(y1, n[5])
                        npx = (n[5] - n[4] + 1) * (n[3] - n[2] + 1); upx = (uy1 - uy0 + 1) * (ux1 - ux0 + 1)
                        if upx * 100 > (spx + npx) * 250 or upx > 2048: break
                        x0, x1, y0, y1 = ux0, ux1, uy0, uy1; cnt += 1; spx += npx
                    W = SHAPES[a[1]][1]
                    xa0, xa1 = x0 & ~3, (x1 | 3)
                    for row in range(y0, y1 + 1):
                        clus[K] += len(lines_of(row, xa0, min(xa1, W - 1), W))
                    k += cnt
        # band sweeps: RoI -> band of its window's CENTRE row (what the visiting order of fpn.hip does); the band stages the rows
        # [min window top, max window bottom] over the x range its windows cover
        for H in bands:
            for lvl, (Hl, Wl) in enumerate(SHAPES):
                lw = [w for w in wins if w[1] == lvl]
                if not lw: continue
                nb = -(-Hl // H)
                top, bottom = [1 << 30] * nb, [-1] * nb
                xs = [[Wl, -1] for _ in range(nb)]
                for w in lw:
                    b = min(((w[4] + w[5]) // 2) // H, nb - 1)
                    top[b] = min(top[b], w[4]); bottom[b] = max(bottom[b], w[5])
                    xs[b][0] = min(xs[b][0], w[2]); xs[b][1] = max(xs[b][1], w[3])
                for b in range(nb):
                    if bottom[b] < 0: continue
                    band_lds[H] = max(band_lds[H], bottom[b] - top[b] + 1)
                    for row in range(top[b], bottom[b] + 1):
                        bands[H] += len(lines_of(row, xs[b][0], xs[b][1], Wl))
    R = tot["rois"]
    print("%d images, %d RoIs; line fills per (RoI, channel):" % (n_img, R))
    print("  compulsory (each touched line once per image and channel)  %.2f" % (tot["compulsory"] / R))
    print("  one RoI per workgroup                                      %.2f" % (tot["roi"] / R))
    for K, v in clus.items():
        print("  clusters of <= %2d consecutive RoIs (merge 250 %%)            %.2f" % (K, v / R))
    for H, v in bands.items():
        print("  band sweep, %2d-row bands (+ halo; tallest band %3d rows)     %.2f" % (H, band_lds[H], v / R))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 2)


def band_batches(n_img, H=32, Wc=64):
    """Feasibility numbers of the band sweep: rows a band's LDS window must hold, and how many RoIs a batch holds
    when the window is a ring of Wc columns and a batch is a run of consecutive RoIs (x-centre order) whose windows fit in it."""
    rows_hist, batch_sizes, per_band, wide = [], [], [], 0
    for i in range(n_img):
        d = np.load("/tmp/rois_%d.npy" % i)
        wins = [window(r) for r in d]
        for lvl, (Hl, Wl) in enumerate(SHAPES):
            lw = [w for w in wins if w[1] == lvl]
            for b in range(-(-Hl // H)):
                bw = sorted([w for w in lw if min(((w[4] + w[5]) // 2) // H, -(-Hl // H) - 1) == b], key=lambda w: (w[2] + w[3]))
                if not bw: continue
                per_band.append((lvl, len(bw)))
                rows_hist.append((lvl, max(w[5] for w in bw) - min(w[4] for w in bw) + 1))
                k = 0
                while k < len(bw):
                    xa, n = bw[k][2], 0
                    while k + n < len(bw):
                        w = bw[k + n]
                        xa2 = min(xa, w[2])
                        if max(x[3] for x in bw[k:k + n + 1]) - xa2 + 1 > Wc: break
                        xa = xa2; n += 1
                    if n == 0: wide += 1; n = 1          # a single window wider than the ring: needs its own path
                    batch_sizes.append((lvl, n)); k += n
    for lvl in range(4):
        r = [x for l, x in rows_hist if l == lvl]; bs = [x for l, x in batch_sizes if l == lvl]; pb = [x for l, x in per_band if l == lvl]
        if not r: continue
        print("  level %d: bands %3d  RoIs/band mean %5.1f max %3d | window rows mean %4.1f max %3d | RoIs per batch mean %4.1f  (batches with < 4 RoIs: %2.0f %%)"
              % (lvl, len(r), np.mean(pb), max(pb), np.mean(r), max(r), np.mean(bs), 100.0 * np.mean(np.array(bs) < 4)))
    print("  windows wider than the %d-column ring: %d" % (Wc, wide))


if __name__ == "__main__":
    for H, Wc in ((32, 64), (32, 128), (16, 64)):
        print("band sweep feasibility, %d-row bands, ring of %d columns:" % (H, Wc))
        band_batches(int(sys.argv[1]) if len(sys.argv) > 1 else 2, H, Wc)
