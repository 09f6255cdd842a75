"""Build query_amd/libn1k.so (HIP kernels + host engine + C ABI) for gfx950 with hipcc.

In-tree build: the .so travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libn1k.so")
SOURCES = ["n1k_kernels.hip", "n1k_plan.cpp", "n1k_engine.cpp", "n1k_jit.cpp", "n1k_json.cpp"]
HEADERS = ["n1k_types.h", "n1k_device.h", "n1k_tables.h", "n1k_spec.h", "n1k_jit.h", "n1k_kernels.h", "n1k_plan.h", os.path.join("..", "..", "include", "n1k.h")]
ARCH = "gfx950"


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    mt = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS:
        if os.path.getmtime(os.path.join(CSRC, f)) > mt:
            return True
    return False


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
           "-Wall", "-Wno-unused-function", "-o", LIB]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-lhiprtc", "-lrccl", "-ldl", "-pthread"]  # hiprtc: run-time instantiation of the plan-specialised kernel
    # (n1k_jit.cpp); rccl: the multi-GPU exchange behind the ABI (n1k_comm_*, n1k_exchange_*)
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
