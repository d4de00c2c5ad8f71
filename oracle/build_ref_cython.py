"""Cythonize the staged copies of the reference's cython_nms.pyx / cython_bbox.pyx (see oracle/Makefile, target `ref`).

TEST INFRASTRUCTURE ONLY.  Run from oracle/_ref/cybuild; plain -O2, no -march=native / -ffast-math so the float32
arithmetic of /root/reference/lib/utils_cython/cython_nms.pyx:37-203 is preserved exactly.
"""
import numpy as np
from Cython.Build import cythonize
from setuptools import Extension, setup

flags = ["-O2", "-Wno-cpp", "-ffp-contract=off", "-Wno-unused-function"]
exts = [
    Extension("cython_nms", ["cython_nms.pyx"], include_dirs=[np.get_include()], extra_compile_args=flags),
    Extension("cython_bbox", ["cython_bbox.pyx"], include_dirs=[np.get_include()], extra_compile_args=flags),
]
setup(name="detectorch_ref_cython", ext_modules=cythonize(exts, language_level=2))
