"""End-to-end ingestion from raw JSON documents (SURVEY §8d iii): n1k_push_json = host extraction + H2D + kernels."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import query_amd
from query_amd import plan
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
D = lambda *x: plan.field_path("default", *x)
rng = np.random.default_rng(1)
cat = rng.integers(0, 1000, n)
price = rng.integers(0, 10000, n) / 100.0
user = rng.integers(0, n // 10 + 1, n)
region = rng.integers(0, 64, n)
t0 = time.perf_counter()
docs = [('{"id":"d%d","cat":"cat_%d","price":%s,"user_id":%d,"region_id":%d,"pad":"%s"}' % (i, cat[i], repr(float(price[i])), user[i], region[i], "x" * 64)).encode()
        for i in range(n)]
print("generated %d docs, %.1f MB, %.1f s" % (n, sum(map(len, docs)) / 1e6, time.perf_counter() - t0))
op = query_amd.GpuFilterGroup(plan.filter_group_plan("(50 < %s)" % D("price"), [D("cat")], ["sum(%s)" % D("price")]))
offsets, blob = op._pack_docs(docs)
import ctypes as C
for threads in (1, 4, 16):
    op.set_option("json_threads", threads)
    b = query_amd._ffi.Batch()
    t0 = time.perf_counter()
    op._check(op._lib.n1k_extract_json(op._h, n, offsets.ctypes.data_as(C.POINTER(C.c_uint64)), blob, C.byref(b)))
    dt = time.perf_counter() - t0
    print("extract, %2d threads: %.3f s  %.2f M docs/s  %.0f MB/s" % (threads, dt, n / dt / 1e6, len(blob) / dt / 1e6))
if query_amd.device_count() > 0:
    for _ in range(3):
        op.reopen()
        t0 = time.perf_counter()
        op._check(op._lib.n1k_push_json(op._h, n, offsets.ctypes.data_as(C.POINTER(C.c_uint64)), blob))
        r = op.after_items_raw()
        dt = time.perf_counter() - t0
        print("push_json + finish: %.3f s  %.2f M docs/s (%d groups)" % (dt, n / dt / 1e6, r["ngroups"]))
