"""Offline: LDS cycles of the tap gather (ds_read_b128 on gfx950: 4 fixed lane groups of 16, 64 banks of 4 B -> 16 slots of 16 B; equal
addresses broadcast) for lane <-> bin assignments, cluster kernel (FPN box head) and map-stationary kernel (C4)."""
import sys
import numpy as np
F = np.float32
GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]

def axis(start, binsz, p, i, grid, extent):
    v = F(start) + F(p) * F(binsz)
    v = F(v + F(F(F(i) + F(.5)) * F(binsz)) / F(grid))
    if v <= 0: v = F(0)
    lo = int(v)
    if lo >= extent - 1: lo = hi = extent - 1
    else: hi = lo + 1
    return lo, hi

def cost(addrs):
    cyc = 0
    for g in GROUPS:
        occ = {}
        for ln in g:
            a = addrs[ln]
            if a is None: continue
            occ.setdefault(a & 15, set()).add(a)
        cyc += max([len(v) for v in occ.values()], default=1)
    return cyc

# ---- lane -> bin maps for ONE RoI per wave (64 lanes, 49 bins): list of 64 entries (ph, pw) or None
def map_rowmajor():
    return [(l // 7, l % 7) if l < 49 else None for l in range(64)]
def map_blocks44():
    m = [None] * 64
    blocks = [[(a, b) for a in range(0, 4) for b in range(0, 4)], [(a, b) for a in range(0, 4) for b in range(4, 7)],
              [(a, b) for a in range(4, 7) for b in range(0, 4)], [(a, b) for a in range(4, 7) for b in range(4, 7)]]
    for g, bl in zip(GROUPS, blocks):
        for ln, pb in zip(g, bl): m[ln] = pb
    return m
def map_rows2():
    m = [None] * 64
    blocks = [[(a, b) for a in (0, 1) for b in range(7)], [(a, b) for a in (2, 3) for b in range(7)], [(a, b) for a in (4, 5) for b in range(7)], [(6, b) for b in range(7)]]
    for g, bl in zip(GROUPS, blocks):
        for ln, pb in zip(g, bl): m[ln] = pb
    return m
def map_cols2():
    return [None if e is None else (e[1], e[0]) for e in map_rows2()]
def map_rows2_interleaved():   # rows (0,4) (1,5) (2,6) (3)
    m = [None] * 64
    blocks = [[(a, b) for a in (0, 4) for b in range(7)], [(a, b) for a in (1, 5) for b in range(7)], [(a, b) for a in (2, 6) for b in range(7)], [(3, b) for b in range(7)]]
    for g, bl in zip(GROUPS, blocks):
        for ln, pb in zip(g, bl): m[ln] = pb
    return m

ONE_ROI_MAPS = {"row-major (49 of 64)": map_rowmajor, "4x4 bin blocks per lane group": map_blocks44, "2 bin rows per lane group": map_rows2, "2 bin cols per lane group": map_cols2,
                "bin rows (k, k+4) per lane group": map_rows2_interleaved}

def fpn_cluster_sim(pitch_mode="kernel"):
    shapes = [(200, 336), (100, 168), (50, 84), (25, 42)]
    scales = [1 / 4., 1 / 8., 1 / 16., 1 / 32.]
    rois = np.load("/tmp/sim/rois_0.npy")
    lv = rois[:, 5].astype(int); fs = lv + 2
    yc = ((rois[:, 2] + rois[:, 4]) * 0.5).astype(np.int64); xc = ((rois[:, 1] + rois[:, 3]) * 0.5).astype(np.int64)
    band = (yc >> fs) >> 4
    order = np.lexsort((np.arange(len(rois)), xc >> fs, band, lv))
    rois = rois[order]
    T = []
    for r in rois:
        l = int(r[5]); H, W = shapes[l]; s = F(scales[l])
        sw, sh = F(r[1]) * s, F(r[2]) * s
        rw = max(F(F(r[3]) * s - sw), F(1)); rh = max(F(F(r[4]) * s - sh), F(1))
        bh, bw = F(rh / F(7)), F(rw / F(7))
        ys = [[axis(sh, bh, p, i, 2, H) for i in range(2)] for p in range(7)]
        xs = [[axis(sw, bw, p, i, 2, W) for i in range(2)] for p in range(7)]
        T.append((l, ys, xs))
    win_bytes = 52 * 1024 - 2576 - 5 * 49 * 16 * 4
    def clusters(K):
        out = []
        for b0 in range(0, len(T), K):
            grp = list(range(b0, min(b0 + K, len(T))))
            k = 0
            while k < len(grp):
                a = T[grp[k]]
                x0, x1, y0, y1 = a[2][0][0][0], a[2][6][1][1], a[1][0][0][0], a[1][6][1][1]
                cnt = 1; sum_px = (y1 - y0 + 1) * (x1 - x0 + 1)
                while k + cnt < len(grp):
                    n = T[grp[k + cnt]]
                    if n[0] != a[0]: break
                    nx0, nx1, ny0, ny1 = n[2][0][0][0], n[2][6][1][1], n[1][0][0][0], n[1][6][1][1]
                    ux0, ux1, uy0, uy1 = min(x0, nx0), max(x1, nx1), min(y0, ny0), max(y1, ny1)
                    ungx = (ux1 >> 2) - (ux0 >> 2) + 1; unpos = (uy1 - uy0 + 1) * ungx
                    if unpos > 512 or (4 * unpos + (unpos >> 1) + 1) * 16 > win_bytes: break
                    npx = (ny1 - ny0 + 1) * (nx1 - nx0 + 1); upx = (uy1 - uy0 + 1) * (ux1 - ux0 + 1)
                    if upx * 100 > (sum_px + npx) * 250: break
                    x0, x1, y0, y1 = ux0, ux1, uy0, uy1; cnt += 1; sum_px += npx
                out.append((grp[k:k + cnt], x0, x1, y0, y1))
                k += cnt
        return out
    def run(name, K, wave_items):
        tot_c, n_inst, n_taps = 0, 0, 0
        for members, x0, x1, y0, y1 in clusters(K):
            x0a = x0 & ~3; tw = 4 * ((x1 >> 2) - (x0 >> 2) + 1)
            if pitch_mode == "kernel":
                f = lambda row, col: (lambda px: px + (px >> 3))((row - y0) * tw + (col - x0a))
            else:
                base = tw + (tw >> 3); pitch = ((base - pitch_mode + 15) // 16) * 16 + pitch_mode
                f = lambda row, col: (row - y0) * pitch + (col - x0a) + ((col - x0a) >> 3)
            for wave in wave_items(len(members)):
                for iy in range(2):
                    for ix in range(2):
                        for tap in range(4):
                            addrs = [None] * 64
                            for ln, it in enumerate(wave):
                                if it is None: continue
                                rl, ph, pw = it
                                _, ys, xs = T[members[rl]]
                                ylo, yhi = ys[ph][iy]; xlo, xhi = xs[pw][ix]
                                addrs[ln] = f((ylo, ylo, yhi, yhi)[tap], (xlo, xhi, xlo, xhi)[tap])
                            tot_c += cost(addrs); n_inst += 1
            n_taps += len(members) * 49 * 16
        print("  %-58s %.2f cycles / ds_read_b128, %.3f LDS cycles per (bin, tap)" % (name, tot_c / n_inst, tot_c / n_taps))
    def packed(count):      # kernel today: tid = rl * 49 + bin
        items = [(i // 49, (i % 49) // 7, (i % 49) % 7) for i in range(count * 49)]
        items += [None] * (-len(items) % 64)
        return [items[w:w + 64] for w in range(0, len(items), 64)]
    print("FPN box head, cluster kernel (K RoIs per workgroup), image 0, pitch", pitch_mode)
    run("K=5: tid = rl*49 + bin  [kernel]", 5, packed)
    for nm, mf in ONE_ROI_MAPS.items():
        m = mf()
        run("K=4: one RoI per wave, " + nm, 4, lambda count, m=m: [[None if e is None else (rl, e[0], e[1]) for e in m] for rl in range(count)])

def c4_sim():
    H, W, S = 50, 84, F(1 / 16.)
    rois = np.load("/tmp/sim/c4_rois.npy")[:300]
    print("C4 map kernel, 300 RoIs of the cfg2 proposals")
    for pitch in (85, 84, 87, 89, 91):
        for nm, mf in ONE_ROI_MAPS.items():
            m = mf()
            tot, n = 0, 0
            for (x1, y1, x2, y2) in rois:
                sw, sh = F(x1) * S, F(y1) * S
                rw = max(F(F(x2) * S - sw), F(1)); rh = max(F(F(y2) * S - sh), F(1))
                bh, bw = F(rh / F(7)), F(rw / F(7))
                gh, gw = int(np.ceil(rh / 7)), int(np.ceil(rw / 7))
                ys = [[axis(sh, bh, p, i, gh, H) for i in range(gh)] for p in range(7)]
                xs = [[axis(sw, bw, p, i, gw, W) for i in range(gw)] for p in range(7)]
                for iy in range(gh):
                    for ix in range(gw):
                        for tap in range(4):
                            addrs = [None] * 64
                            for ln, e in enumerate(m):
                                if e is None: continue
                                ph, pw = e
                                ylo, yhi = ys[ph][iy]; xlo, xhi = xs[pw][ix]
                                addrs[ln] = (ylo, ylo, yhi, yhi)[tap] * pitch + (xlo, xhi, xlo, xhi)[tap]
                            tot += cost(addrs); n += 1
            print("  pitch %d  %-40s %.2f cycles / ds_read_b128" % (pitch, nm, tot / n))

if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "mask"):
    if len(sys.argv) > 1 and sys.argv[1] == "c4": c4_sim()
    else:
        fpn_cluster_sim("kernel")
        for sk in (1, 3, 5, 7): fpn_cluster_sim(sk)


def mask_head_sim():
    """14 x 14 bins, one RoI per workgroup (the mask head of the cluster kernel): tid = bin against one bin ROW per 16-lane group."""
    rs = np.random.RandomState(5)
    shapes = [(200, 336), (100, 168), (50, 84), (25, 42)]
    res = {"tid = bin [kernel]": [0, 0], "one bin row per lane group (14 of 16 lanes)": [0, 0]}
    for _ in range(150):
        side = np.exp(rs.uniform(np.log(40), np.log(700))); ar = np.exp(rs.uniform(-0.6, 0.6))
        w, h = side * np.sqrt(ar), side / np.sqrt(ar)
        lvl = int(np.clip(np.floor(4 + np.log2(np.sqrt(w * h) / 224 + 1e-6)), 2, 5)) - 2
        H, W = shapes[lvl]; s = F(1.0 / (4 << lvl))
        cx, cy = rs.uniform(w / 2, 1333 - w / 2), rs.uniform(h / 2, 800 - h / 2)
        x1, y1, x2, y2 = cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2
        sw, sh = F(x1) * s, F(y1) * s
        rw = max(F(F(x2) * s - sw), F(1)); rh = max(F(F(y2) * s - sh), F(1))
        bh, bw = F(rh / F(14)), F(rw / F(14))
        ys = [[axis(sh, bh, p, i, 2, H) for i in range(2)] for p in range(14)]
        xs = [[axis(sw, bw, p, i, 2, W) for i in range(2)] for p in range(14)]
        x0, x1i, y0 = xs[0][0][0], xs[13][1][1], ys[0][0][0]
        x0a = x0 & ~3; tw = 4 * ((x1i >> 2) - (x0 >> 2) + 1)
        f = lambda row, col: (lambda px: px + (px >> 3))((row - y0) * tw + (col - x0a))
        maps = {}
        maps["tid = bin [kernel]"] = [[(b // 14, b % 14) if b < 196 else None for b in range(wv * 64, wv * 64 + 64)] for wv in range(4)]
        m2 = []
        for wv in range(4):
            lanes = [None] * 64
            for gi, g in enumerate(GROUPS):
                row = wv * 4 + gi
                for pos, ln in enumerate(g):
                    if row < 14 and pos < 14: lanes[ln] = (row, pos)
            m2.append(lanes)
        maps["one bin row per lane group (14 of 16 lanes)"] = m2
        for nm, waves in maps.items():
            for wave in waves:
                if all(e is None for e in wave): continue
                for iy in range(2):
                    for ix in range(2):
                        for tap in range(4):
                            addrs = [None] * 64
                            for ln, e in enumerate(wave):
                                if e is None: continue
                                ph, pw = e
                                ylo, yhi = ys[ph][iy]; xlo, xhi = xs[pw][ix]
                                addrs[ln] = f((ylo, ylo, yhi, yhi)[tap], (xlo, xhi, xlo, xhi)[tap])
                            res[nm][0] += cost(addrs); res[nm][1] += 1
    for nm, (c, n) in res.items():
        print("mask head 14 x 14: %-46s %.2f LDS cycles per ds_read_b128 wave-instruction (%d instructions per RoI and tap set)" % (nm, c / n, n // 150))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "mask":
    mask_head_sim()
