"""Host-buffer ingestion (n1k_push_batch): PCIe-inclusive rate of config 2's query (SURVEY §8d ii)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench, query_amd
from query_amd import plan, _ffi
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
batch_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
D = bench.D
cols = bench.DeviceColumns(rows, 1000, False, 0, rows, 0)
host = {"cat": cols.cat.cpu().numpy().view(np.uint32), "pt": cols.price_t.cpu().numpy(), "pp": cols.price_p.cpu().numpy().view(np.uint64)}


class Col:
    def __init__(self, kind, tags=None, payload=None, codes=None):
        self.kind, self.tags, self.payload, self.codes = kind, tags, payload, codes


op = query_amd.GpuFilterGroup(plan.filter_group_plan("(50 < %s)" % D("price"), [D("cat")], ["sum(%s)" % D("price")]))
op.intern(bench.synth_dictionary(1000))
assert op.column_paths == [D("price"), D("cat")]
for it in range(3):
    op.reopen()
    t0 = time.perf_counter()
    for lo in range(0, rows, batch_rows):
        hi = min(rows, lo + batch_rows)
        op.process_items([Col(_ffi.COL_TAGGED64, tags=host["pt"][lo:hi], payload=host["pp"][lo:hi]),
                              Col(_ffi.COL_DICT32, codes=host["cat"][lo:hi])])
    r = op.after_items_raw()
    dt = time.perf_counter() - t0
    print("push_batch x%d + finish: %.3f s  %.2f G rows/s  %.1f GB/s (13 B/row), %d groups" %
          ((rows + batch_rows - 1) // batch_rows, dt, rows / dt / 1e9, rows * 13 / dt / 1e9, r["ngroups"]))
