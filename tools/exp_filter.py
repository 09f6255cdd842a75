"""Filter-only path (execution/filter.go): mask + scan + compaction kernels and the copy of the survivors' ordinals."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, query_amd
from query_amd import plan
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
D = bench.D
cols = bench.DeviceColumns(rows, 1000, False, 0, rows, 0)
op = query_amd.GpuFilterGroup(plan.filter_group_plan("(50 < %s)" % D("price"), [], [], filter_only=True))
op.intern(bench.synth_dictionary(1000))
batch = op.make_device_batch(rows, [cols.by_path[p] for p in op.column_paths])
for i in range(5):
    op.reopen()
    t0 = time.perf_counter()
    op.process_device_batch(batch)
    op.sync()
    t1 = time.perf_counter()
    r = op.after_items_raw()
    t2 = time.perf_counter()
    print("push+kernels %.3f ms (device %.3f ms), finish %.3f ms, selected %d" % ((t1 - t0) * 1e3, op.stats()["device_ms"], (t2 - t1) * 1e3, len(r["selected"])))
