import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from oracle import n1o
from query_amd import distributed as qd, plan, _ffi
import test_gpu_distributed as T
n = 120_000
t = n1o.synth_table(n, k_cat=5000)
op = qd.ShardedFilterGroup(T.COND, T.KEYS, T.AGGS, t.dictionary, 0, 1, 0)
dev, keep = T._device_cols(t, op.send_paths)
orig = op._gather
def spy(rcv, raw):
    k = raw["keys"]["v"][:, 0]
    print("local groups", raw["ngroups"], "unique", len(np.unique(k)))
    out = orig(rcv, raw)
    k2 = out["keys"]["v"][:, 0]
    print("gathered groups", out["ngroups"], "unique", len(np.unique(k2)), "tags", np.unique(out["keys"]["tag"]))
    print(k[:8], k2[:8])
    return out
op._gather = spy
raw, info = op.run_partials(n, dev)
print(info)
op2 = qd.ShardedFilterGroup(T.COND, T.KEYS, T.AGGS, t.dictionary, 0, 1, 0)
raw2, info2 = op2.run_gathered(n, dev)
k = raw2["keys"]["v"][:, 0]
print("gathered mode: groups", raw2["ngroups"], "unique", len(np.unique(k)), info2)
# plain single-handle run for reference
import query_amd
h = query_amd.GpuFilterGroup(plan.filter_group_plan(T.COND, T.KEYS, T.AGGS)); h.intern(list(t.dictionary))
h.process_device_items(n, [dev[p] for p in h.column_paths]); r = h.after_items_raw()
print("single:", r["ngroups"], len(np.unique(r["keys"]["v"][:,0])))
