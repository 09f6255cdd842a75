"""Where a step's time goes on the HOST (calibration, not product): config 2 over resident columns, N1K_HOST_TRACE=1 for the
library's own split; here the split between the ABI call and the Python conversion of the result."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("N1K_HOST_TRACE", "1")
import torch  # noqa: E402
import bench  # noqa: E402
import query_amd  # noqa: E402
from query_amd import _ffi  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
steps = 300
wl = bench.workloads()["config2"]
cols = bench.DeviceColumns(rows, 1000, False, 0, rows, 0)
op = query_amd.GpuFilterGroup(query_amd.plan.filter_group_plan(wl["cond"], wl["keys"], wl["aggs"]))
for o in sys.argv[2:]:
    k, v = o.split("=")
    op.set_option(k, int(v))
op.intern(bench.synth_dictionary(1000))
batch = op.make_device_batch(rows, [cols.by_path[p] for p in op.column_paths])
for _ in range(20):
    op.run_device_batch_raw(batch)
torch.cuda.synchronize()
res = _ffi.Result()
t_call = t_conv = 0.0
t0 = time.perf_counter()
for _ in range(steps):
    a = time.perf_counter()
    st = op._lib.n1k_run_device_batch(op._h, C.byref(batch[0]), C.byref(res))
    b = time.perf_counter()
    op.after_items_raw(res)
    c = time.perf_counter()
    t_call += b - a
    t_conv += c - b
tot = time.perf_counter() - t0
s = op.stats()
print("rows %d: %.1f us per step = ABI call %.1f us + python conversion %.1f us; device query_ms %.1f us, batch kernels %.1f us" %
      (rows, tot / steps * 1e6, t_call / steps * 1e6, t_conv / steps * 1e6, s["query_ms"] * 1e3, s["device_ms"] * 1e3))
op.done()
