"""krep_amd — MI355X-native literal-scan backend behind krep's search_func_t boundary.

The product is the C-ABI shared library krep_amd/lib/libkrep_gpu.so (include/krep_gpu.h), built from
the hand-written HIP sources in krep_amd/csrc.  This Python package is only the ctypes binding used by
tests/ and bench.py; it raises loudly when the library is missing (no CPU fallback).
"""
from .abi import (Params, make_params, SIZE_MAX, REF_SCALAR, REF_SSE42, REF_AVX2, REF_AVX512)  # noqa: F401
from .engine import Engine, load, KrepGpuError  # noqa: F401
