"""The C ABI without Python or torch: tests/cxx/abi_harness.cpp allocates with the HIP runtime, calls libdetectorch_hip.so
through include/detectorch_hip.h (the reference-shaped launch_roi_align_forward_hip, dtc_roi_align_forward, dtc_nms,
dtc_bbox_overlaps) and compares bit-for-bit with liboracle.so.  Compiled with hipcc on the GPU box.  -m gpu."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_harness_bit_exact(tmp_path, oracle):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    libdir, orcdir = os.path.join(ROOT, "detectorch_amd", "lib"), os.path.join(ROOT, "oracle")
    assert os.path.exists(os.path.join(libdir, "libdetectorch_hip.so")), "native library missing"
    oracle.lib()                                                       # builds oracle/liboracle.so if needed
    exe = str(tmp_path / "abi_harness")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", os.path.join(ROOT, "tests", "cxx", "abi_harness.cpp"),
                           "-I" + os.path.join(ROOT, "include"), "-L" + libdir, "-ldetectorch_hip", "-L" + orcdir, "-loracle",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath," + orcdir, "-o", exe])
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":" + orcdir + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=120)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all checks bit-exact" in out.stdout
