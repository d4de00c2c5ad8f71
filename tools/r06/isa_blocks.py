"""Static instruction mix of a kernel's basic blocks (hipcc -S): per block the number of vector / LDS / VMEM / scalar instructions and
its commonest opcodes -- a quick way to see what a pass of a hot loop really costs (round 6 found 72 if-converted v_cndmask in the
cluster kernel's LDS commit and six quarter-rate 64-bit multiplies per slab store this way: tools/r06/README.md 7).

    python tools/r06/isa_blocks.py detectorch_amd/csrc/roi_align_tile.hip 'roi_align_fwd_tileI6__halfS1_Li256ELb1' [min_valu]

The second argument is a substring of the MANGLED kernel name (list them with: ... | grep '^_ZN3dtc.*:$')."""
import os
import re
import subprocess
import sys
import tempfile

src, pat = sys.argv[1], sys.argv[2]
min_valu = int(sys.argv[3]) if len(sys.argv) > 3 else 25
out = os.path.join(tempfile.mkdtemp(prefix="isa_"), "k.s")
flags = "--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wno-unused-function -S --cuda-device-only".split()
subprocess.run(["/opt/rocm/bin/hipcc"] + flags + [src, "-o", out], check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l and l.rstrip().endswith(":") or (l.startswith("_Z") and pat in l and "; @" in l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks, cur = [], {"name": "(entry)", "line": start, "ops": {}}
blocks.append(cur)
for i in range(start + 1, end + 1):
    l = lines[i]
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = {"name": m.group(1), "line": i - start, "ops": {}, "loop": "Loop" in l}
        blocks.append(cur)
        continue
    t = l.strip().split(" ")[0]
    if t and not t.startswith(";") and not t.startswith("."):
        cur["ops"][t] = cur["ops"].get(t, 0) + 1
cls = lambda b, p: sum(c for o, c in b["ops"].items() if o.startswith(p))
print("%-12s %6s %5s %4s %5s %5s  commonest" % ("block", "line", "valu", "lds", "vmem", "salu"))
for b in blocks:
    v, d, g, sc = cls(b, "v_"), cls(b, "ds_"), cls(b, "global_") + cls(b, "buffer_"), cls(b, "s_")
    if v >= min_valu or d >= 8 or g >= 8:
        top = ", ".join("%s x%d" % kv for kv in sorted(b["ops"].items(), key=lambda x: -x[1])[:6])
        print("%-12s %6d %5d %4d %5d %5d  %s" % (b["name"], b["line"], v, d, g, sc, top))
print("total static: valu %d, lds %d, vmem %d" % (sum(cls(b, "v_") for b in blocks), sum(cls(b, "ds_") for b in blocks),
                                                  sum(cls(b, "global_") + cls(b, "buffer_") for b in blocks)))
