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
SOURCES = ["n1k_kernels.hip", "n1k_bins.hip", "n1k_jsondev.hip", "n1k_jsonpush.cpp", "n1k_plan.cpp", "n1k_engine.cpp", "n1k_scan.cpp", "n1k_partitioned.cpp", "n1k_distinct.cpp",
           "n1k_finish.cpp", "n1k_tail.cpp", "n1k_exchange.cpp", "n1k_jit.cpp", "n1k_json.cpp"]
HEADERS = ["n1k_types.h", "n1k_device.h", "n1k_tables.h", "n1k_scatter.h", "n1k_spec.h", "n1k_jit.h", "n1k_kernels.h", "n1k_plan.h", os.path.join("..", "..", "include", "n1k.h")]
HOST_HEADERS = ["n1k_engine.h"]  # host-only: not part of source_hash()
ARCH = "gfx950"


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


OBJDIR = os.path.join(CSRC, "build")  # objects: git-ignored, rebuilt per source when it or any header is newer
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    mt = os.path.getmtime(target)
    return any(os.path.getmtime(d) > mt for d in deps)


def source_hash() -> str:
    """sha1 over the DEVICE code (the two kernel translation units and every header they include): stamps measurements
    (profiles/*_pmc_*.json) with the kernels they belong to."""
    import hashlib
    h = hashlib.sha1()
    for f in sorted([x for x in SOURCES if x.endswith(".hip")] + [x for x in HEADERS if not x.startswith("..")]):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def needs_build() -> bool:
    return _stale(LIB, [os.path.join(CSRC, f) for f in SOURCES + HEADERS + HOST_HEADERS] + [os.path.abspath(__file__)])


def build(force: bool = False, verbose: bool = False) -> str:
    """One object per source (compiled in parallel, only the stale ones), then one link."""
    if not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in HEADERS + HOST_HEADERS] + [os.path.abspath(__file__)]

    def compile_one(src: str) -> str:
        obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")
        path = os.path.join(CSRC, src)
        deps = headers if not src.endswith(".hip") else [os.path.join(CSRC, f) for f in HEADERS]  # (kernels see no host header)
        if force or _stale(obj, [path] + deps):
            cmd = [hipcc] + FLAGS + ["-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    # hiprtc: run-time instantiation of the plan-specialised kernel (n1k_jit.cpp); rccl: the multi-GPU exchange behind
    # the ABI (n1k_comm_*, n1k_exchange_*)
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-lhiprtc", "-lrccl", "-ldl", "-pthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
