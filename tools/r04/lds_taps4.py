"""Offline: LDS cycles per ds_read_b128 of the cluster kernel's tap gather (CDNA4: lane groups of 16, 16 slots of 16 B; equal
addresses inside a group broadcast) for other lane <-> (bin, sample, tap) mappings and row pitches, on the bench RoIs in the
kernel's visiting order (/tmp/rois_0.npy written by tools/r03/band_bench.py).  4.0 = conflict-free.

  python tools/r04/lds_taps4.py
"""
import sys
import numpy as np
sys.path.insert(0, "tools/r03")
from band_model import axis, F
from lds_taps import GROUPS, roi_taps

shapes = [(200, 336), (100, 168), (50, 84), (25, 42)]
scales = [1 / 4., 1 / 8., 1 / 16., 1 / 32.]


def cost(addr_lists):
    """addr_lists: per lane an address (slot) or None -> cycles of one wave instruction"""
    cyc = 0
    for g in GROUPS:
        occ = {}
        for ln in g:
            a = addr_lists[ln]
            if a is None: continue
            occ.setdefault(a & 15, set()).add(a)
        cyc += max([len(v) for v in occ.values()], default=1)
    return cyc


def main():
    rois = np.load("/tmp/rois_0.npy")
    taps = [roi_taps(r, shapes, scales) for r in rois[:400]]
    pitches = {"kernel (px + px/8)": None, "pitch 16k+1": 1, "pitch 16k+5": 5, "pitch 16k+7": 7, "pitch 16k+9": 9}
    maps = ["lane=bin, instr=(sample,tap)  [kernel]", "lane=(bin,sample) s=lane&3, instr=(iter,tap)", "lane=(bin,sample) s=lane>>4&3, instr=(iter,tap)",
            "lane=(bin,tap) t=lane&3, instr=(iter,sample)", "lane=(bin,tap) t=lane>>4&3, instr=(iter,sample)"]
    for pname, skew in pitches.items():
        for mname in maps:
            tot, n = 0, 0
            for b0 in range(0, 400, 5):
                grp = list(range(b0, b0 + 5))
                x0 = min(taps[k][1][0][0][0] for k in grp) & ~3
                x1 = max(taps[k][1][6][1][1] for k in grp)
                y0 = min(taps[k][0][0][0][0] for k in grp)
                tw = 4 * ((x1 >> 2) - (x0 >> 2) + 1)
                if skew is None:
                    f = lambda row, col: (lambda px: px + (px >> 3))((row - y0) * tw + (col - x0))
                else:
                    base = tw + (tw >> 3)
                    pitch = ((base - skew + 15) // 16) * 16 + skew
                    f = lambda row, col: (row - y0) * pitch + (col - x0) + ((col - x0) >> 3)

                def addr(k, bn, iy, ix, tap):
                    ph, pw = divmod(bn, 7)
                    ylo, yhi = taps[k][0][ph][iy]; xlo, xhi = taps[k][1][pw][ix]
                    return f((ylo, ylo, yhi, yhi)[tap], (xlo, xhi, xlo, xhi)[tap])
                items = [(k, bn) for k in grp for bn in range(49)]          # 245 items
                if mname.startswith("lane=bin"):
                    for w0 in range(0, 256, 64):
                        for s in range(4):
                            for tap in range(4):
                                a = [addr(*items[w0 + ln], s >> 1, s & 1, tap) if w0 + ln < len(items) else None for ln in range(64)]
                                tot += cost(a); n += 1
                else:
                    sub_is_sample = "sample)" in mname.split(",")[1] if False else mname.startswith("lane=(bin,sample)")
                    low = "lane&3" in mname
                    # 256 lanes x 4 iterations cover 245 x 4 (item, sub); wave w, iteration j: items 16 * (4 * j + w) ... + 15
                    for j in range(4):
                        for w in range(4):
                            for other in range(4):
                                a = []
                                for ln in range(64):
                                    if low: sub, il = ln & 3, ln >> 2
                                    else: sub, il = (ln >> 4) & 3, ln & 15
                                    i = 16 * (4 * j + w) + il
                                    if i >= len(items): a.append(None); continue
                                    s, tap = (sub, other) if sub_is_sample else (other, sub)
                                    a.append(addr(*items[i], s >> 1, s & 1, tap))
                                tot += cost(a); n += 1
            print("%-22s %-52s %.2f cycles per ds_read_b128, %d reads per 5-RoI group and quad" % (pname, mname, tot / n, n // 80))


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def qlane():
    """lane <-> (channel quad q = lane & 3, bin slot): a 16-lane group reads 4 bins x 4 resident quads; slot = q * plane + phys(px)"""
    rois = np.load("/tmp/rois_0.npy")
    taps = [roi_taps(r, shapes, scales) for r in rois[:400]]
    for res in (None, 1, 4, 5, 7, 12):
        for order in ("q=lane&3", "q=lane>>4&3"):
            tot, n = 0, 0
            for b0 in range(0, 400, 5):
                grp = list(range(b0, b0 + 5))
                x0 = min(taps[k][1][0][0][0] for k in grp) & ~3
                x1 = max(taps[k][1][6][1][1] for k in grp)
                y0 = min(taps[k][0][0][0][0] for k in grp)
                y1 = max(taps[k][0][6][1][1] for k in grp)
                tw = 4 * ((x1 >> 2) - (x0 >> 2) + 1)
                npos = (y1 - y0 + 1) * (tw // 4)
                plane = 4 * npos + (npos >> 1) + 1
                if res is not None:
                    plane = ((plane - res + 15) // 16) * 16 + res
                f = lambda row, col: (lambda px: px + (px >> 3))((row - y0) * tw + (col - x0))
                items = [(k, bn) for k in grp for bn in range(49)]
                # a wave-instruction: 16 bins x 4 quads; 245 bins -> 16 instruction groups (4 waves x 4 iterations)
                for w0 in range(0, 256, 16):
                    for s in range(4):
                        for tap in range(4):
                            a = []
                            for ln in range(64):
                                if order == "q=lane&3": q, il = ln & 3, ln >> 2
                                else: q, il = (ln >> 4) & 3, ln & 15
                                i = w0 + il
                                if i >= len(items): a.append(None); continue
                                k, bn = items[i]
                                ph, pw = divmod(bn, 7)
                                ylo, yhi = taps[k][0][ph][s >> 1]; xlo, xhi = taps[k][1][pw][s & 1]
                                a.append(q * plane + f((ylo, ylo, yhi, yhi)[tap], (xlo, xhi, xlo, xhi)[tap]))
                            tot += cost(a); n += 1
            print("plane %% 16 = %-6s %-12s %.2f cycles per ds_read_b128 (kernel today 9.60)" % (res, order, tot / n))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "qlane":
    qlane()


def b64():
    """16-bit LDS image: a tap is ds_read_b64 (4 channels x 2 B), 32 lanes per pass over 32 slots of 8 B (2 cycles conflict-free in the
    units in which a conflict-free ds_read_b128 costs 4)"""
    rois = np.load("/tmp/rois_0.npy")
    taps = [roi_taps(r, shapes, scales) for r in rois[:400]]
    for pad in ("px + px/8", "px + px/16", "none"):
        tot, n = 0, 0
        for b0 in range(0, 400, 5):
            grp = list(range(b0, b0 + 5))
            x0 = min(taps[k][1][0][0][0] for k in grp) & ~3
            x1 = max(taps[k][1][6][1][1] for k in grp)
            y0 = min(taps[k][0][0][0][0] for k in grp)
            tw = 4 * ((x1 >> 2) - (x0 >> 2) + 1)
            def f(row, col):
                px = (row - y0) * tw + (col - x0)
                return px + (px >> 3) if pad == "px + px/8" else px + (px >> 4) if pad == "px + px/16" else px
            items = [(k, bn) for k in grp for bn in range(49)]
            for w0 in range(0, 256, 64):
                for s in range(4):
                    for tap in range(4):
                        cyc = 0
                        for half in (range(0, 32), range(32, 64)):
                            occ = {}
                            for ln in half:
                                if w0 + ln >= len(items): continue
                                k, bn = items[w0 + ln]
                                ph, pw = divmod(bn, 7)
                                ylo, yhi = taps[k][0][ph][s >> 1]; xlo, xhi = taps[k][1][pw][s & 1]
                                a = f((ylo, ylo, yhi, yhi)[tap], (xlo, xhi, xlo, xhi)[tap])
                                occ.setdefault(a & 31, set()).add(a)
                            cyc += max([len(v) for v in occ.values()], default=1)
                        tot += cyc; n += 1
        print("16-bit image, pad %-12s %.2f cycles per ds_read_b64 (conflict-free 2; today's ds_read_b128 on the float32 image: 9.60)" % (pad, tot / n))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "b64":
    b64()
