"""Offline: LDS cycles per ds_read_b128 of the map-stationary kernel's tap gather (C4: one 50 x 84 map, lane <-> bin of one RoI,
adaptive sampling grid) for row pitches of the LDS image (slots of 16 B; the kernel today: pitch = W = 84).
RoIs: anchors of 32..512 px x 3 aspect ratios at random positions, clipped to 1333 x 800 (what RPN on noise proposes)."""
import sys
import numpy as np
import numpy.random          # before tools/r03 enters the path: its bisect.py would shadow the standard module
sys.path.insert(0, "tools/r03")
from band_model import axis, F
from lds_taps import GROUPS

H, W, S = 50, 84, F(1 / 16.)
rs = np.random.RandomState(3)


def rois(n):
    out = []
    for _ in range(n):
        sc = rs.choice([32, 64, 128, 256, 512]) * np.exp(rs.normal(0, 0.25)); ar = rs.choice([0.5, 1.0, 2.0])
        w, h = sc * np.sqrt(1 / ar), sc * np.sqrt(ar)
        cx, cy = rs.uniform(0, 1333), rs.uniform(0, 800)
        out.append((max(cx - w / 2, 0), max(cy - h / 2, 0), min(cx + w / 2, 1332), min(cy + h / 2, 799)))
    return out


def main():
    R = rois(300)
    for pitch in (84, 85, 87, 89, 91, 93):
        tot, n = 0, 0
        for (x1, y1, x2, y2) in R:
            sw, sh = F(x1) * S, F(y1) * S
            rw = max(F(F(x2) * S - sw), F(1)); rh = max(F(F(y2) * S - sh), F(1))
            bh, bw = F(rh / F(7)), F(rw / F(7))
            gh, gw = int(np.ceil(rh / 7)), int(np.ceil(rw / 7))
            ys = [[axis(sh, bh, p, i, gh, H) for i in range(gh)] for p in range(7)]
            xs = [[axis(sw, bw, p, i, gw, W) for i in range(gw)] for p in range(7)]
            for iy in range(gh):
                for ix in range(gw):
                    for tap in range(4):
                        cyc = 0
                        for g in GROUPS:
                            occ = {}
                            for ln in g:
                                if ln >= 49: continue
                                ph, pw = divmod(ln, 7)
                                ylo, yhi = ys[ph][iy]; xlo, xhi = xs[pw][ix]
                                a = (ylo, ylo, yhi, yhi)[tap] * pitch + (xlo, xhi, xlo, xhi)[tap]
                                occ.setdefault(a & 15, set()).add(a)
                            cyc += max([len(v) for v in occ.values()], default=1)
                        tot += cyc; n += 1
        print("row pitch %d slots: %.2f cycles per ds_read_b128 (4 = conflict-free), %d reads" % (pitch, tot / n, n))


if __name__ == "__main__":
    main()
