"""query_amd — MI355X-native Filter -> Group -> Aggregate for the N1QL engine.

The product is query_amd/libn1k.so (hand-written HIP kernels for gfx950 + a
C++ host engine behind the C ABI of include/n1k.h).  This package is the thin
Python host layer used by tests, bench.py and the multi-GPU driver; it holds
no compute of its own and no CPU fallback.
"""
from . import _ffi, plan  # noqa: F401
from .gpu_operator import GpuFilterGroup, GroupRows, N1kError, device_count  # noqa: F401

__all__ = ["GpuFilterGroup", "GroupRows", "N1kError", "device_count", "plan"]
