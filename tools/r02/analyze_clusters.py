"""Offline model of roi_align_fwd_tile's clustering on real descriptors (development aid)."""
import sys, numpy as np
SCALES = (0.25, 0.125, 0.0625, 0.03125); SHAPES = ((200, 336), (100, 168), (50, 84), (25, 42))

def axis(start, binsz, p, i, grid, extent):
    v = np.float32(start) + np.float32(p) * np.float32(binsz)
    v = np.float32(v + np.float32((np.float32(i) + np.float32(.5)) * np.float32(binsz)) / np.float32(grid))
    if v <= 0: v = np.float32(0)
    lo = int(v)
    if lo >= extent - 1: lo = hi = extent - 1
    else: hi = lo + 1
    return lo, hi

def window(d, P=7):
    b, x1, y1, x2, y2, lvl = d[:6]
    lvl = int(lvl)
    if lvl < 0: return None
    s = np.float32(SCALES[lvl]); H, W = SHAPES[lvl]
    sw, sh = np.float32(x1) * s, np.float32(y1) * s
    rw, rh = max(np.float32(x2) * s - sw, np.float32(1)), max(np.float32(y2) * s - sh, np.float32(1))
    bh, bw = rh / np.float32(P), rw / np.float32(P)
    return (int(b), lvl, axis(sw, bw, 0, 0, 2, W)[0], axis(sw, bw, P - 1, 1, 2, W)[1], axis(sh, bh, 0, 0, 2, H)[0], axis(sh, bh, P - 1, 1, 2, H)[1])

def run(desc, K, NT, lds_kb, merge_pct, nq_cap=2, U=8, bins=49, cb=64):
    NW = NT // 64; max_pos = U * NW * 16
    win_bytes = lds_kb * 1024 - 2064 - K * bins * 16 * nq_cap
    R = desc.shape[0]
    wins = [window(d) for d in desc]
    ncl = 0; passes = 0; px_union = 0; px_sum = 0; fills = 0; items = 0; counts = []
    for g0 in range(0, R, K):
        k = 0; grp = wins[g0:g0 + K]
        while k < len(grp):
            a = grp[k]
            if a is None: k += 1; continue
            x0, x1, y0, y1 = a[2], a[3], a[4], a[5]; cnt = 1
            spx = (y1 - y0 + 1) * (x1 - x0 + 1)
            while k + cnt < len(grp) and (cnt + 1) * bins <= NT:
                n = grp[k + cnt]
                if n is None or n[0] != a[0] or n[1] != a[1]: break
                ux0, ux1, uy0, uy1 = min(x0, n[2]), max(x1, n[3]), min(y0, n[4]), max(y1, n[5])
                ungx = (ux1 >> 2) - (ux0 >> 2) + 1; unpos = (uy1 - uy0 + 1) * ungx
                if unpos > max_pos or (4 * unpos + (unpos >> 1) + 1) * 16 > win_bytes: break
                npx = (n[5] - n[4] + 1) * (n[3] - n[2] + 1); upx = (uy1 - uy0 + 1) * (ux1 - ux0 + 1)
                if upx * 100 > (spx + npx) * merge_pct: break
                x0, x1, y0, y1 = ux0, ux1, uy0, uy1; cnt += 1; spx += npx
            ngx = (x1 >> 2) - (x0 >> 2) + 1; npos = (y1 - y0 + 1) * ngx
            plane = 4 * npos + (npos >> 1) + 1
            KC = -(-((npos + 15) >> 4) // NW)
            nqp = max(1, min(win_bytes // (plane * 16), U // KC, cb // 4, nq_cap))
            passes += -(-(cb // 4) // nqp); ncl += 1; counts.append(cnt)
            px_union += npos * 4; px_sum += spx; items += cnt
            esz = 4; W = SHAPES[a[1]][1]
            for row in range(y0, y1 + 1):
                b0 = (row * W + (x0 & ~3)) * esz; b1 = (row * W + (x0 & ~3) + 4 * ngx) * esz - 1
                fills += (b1 >> 7) - (b0 >> 7) + 1
            k += cnt
    ngrp = -(-R // K)
    print("K=%d NT=%d merge=%d: clusters/WG %.2f  RoIs/cluster %.2f  passes/WG(cb=%d) %.1f  staged px/RoI %.0f (sum of windows %.0f)  line fills/(RoI,channel) %.1f"
          % (K, NT, merge_pct, ncl / ngrp, items / ncl, cb, passes / ngrp, px_union / items, px_sum / items, fills / items))
    return counts

if __name__ == "__main__":
    desc = np.load(sys.argv[1]).reshape(-1, 8)
    lv = desc[:, 5].astype(int)
    print("levels:", np.bincount(lv[lv >= 0], minlength=4), "padding:", (lv < 0).sum())
    for (K, NT, lds) in ((1, 256, 52), (5, 256, 52), (10, 512, 78), (20, 1024, 156)):
        for m in (150, 250, 400):
            run(desc, K, NT, lds, m, U=8 if NT == 256 else 4)
