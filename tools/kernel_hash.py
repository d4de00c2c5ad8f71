"""sha256[:16] of a kernel's source with comments and blank space removed (so that a comment edit does not invalidate a committed
counter measurement, and a code edit does).  Used by tools/collect_profiles.sh to stamp profiles/roialign_traffic.json and by
bench.py to decide whether the committed `roofline.traffic` still belongs to the kernel that is running."""
import hashlib
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "detectorch_amd", "csrc")
# the file that holds the dominant RoIAlign kernel of each bench workload (+ the helpers it is built from)
KERNEL_FILES = {"cfg3": ["roi_align_tile.hip", "roi_align_common.h"], "cfg5": ["roi_align_tile.hip", "roi_align_common.h"],
                "cfg2": ["roi_align_map.hip", "roi_align_common.h"]}


def strip(src):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    return re.sub(r"\s+", " ", src).strip()


# channels_last feature maps (bench.py --channels-last) are pooled by the kernels of these files whatever the workload
NHWC_FILES = ["roi_align.hip", "roi_align_nhwc.hip", "roi_align_nhwc16.hip", "roi_align_common.h"]


def kernel_sha16(workload, channels_last=False):
    h = hashlib.sha256()
    for f in (NHWC_FILES if channels_last else KERNEL_FILES[workload]):
        h.update(strip(open(os.path.join(CSRC, f)).read()).encode())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_sha16(sys.argv[1] if len(sys.argv) > 1 else "cfg3", channels_last="nhwc" in sys.argv[2:]))
