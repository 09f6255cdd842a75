import os, sys, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from oracle import n1o
import query_amd
from query_amd import distributed as qd, plan, _ffi
import test_gpu_distributed as T
n = 120_000
t = n1o.synth_table(n, k_cat=5000)
dev_cols = None
for shared in (True, False):
    st = torch.cuda.Stream()
    opts = {"stream": st.cuda_stream} if shared else {}
    snd = query_amd.GpuFilterGroup(plan.filter_group_plan(T.COND, T.KEYS, T.AGGS), device=0, **opts)
    rcv = query_amd.GpuFilterGroup(plan.filter_group_plan(None, T.KEYS, T.AGGS), device=0, **opts)
    snd.intern(list(t.dictionary)); rcv.intern(list(t.dictionary))
    dev, keep = T._device_cols(t, snd.column_paths)
    comm = qd.Comm(0, 1, 0)
    lib = snd._lib
    snd.process_device_items(n, [dev[p] for p in snd.column_paths])
    snd.sync()
    print("sender groups", snd.stats()["groups_out"])
    for gathered in (1, 0):
        rcv.reopen()
        snd._check(lib.n1k_exchange_partials(comm._h, snd._h, rcv._h, 8192, gathered))
        r = rcv.after_items_raw()
        print("shared" if shared else "separate", "gathered" if gathered else "a2a", r["ngroups"], len(np.unique(r["keys"]["v"][:, 0])), np.unique(r["keys"]["tag"]))
