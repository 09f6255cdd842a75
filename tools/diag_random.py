import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import test_gpu_random_plans as T
import parity_util as pu
from oracle import n1o
seed = int(sys.argv[1])
rng = np.random.default_rng(1000 + seed)
t = T.make_table(rng, int(rng.integers(1, 6000)))
cond, keys, aggs = T.rand_plan(rng)
opts = dict(T.OPTION_SETS[rng.integers(0, len(T.OPTION_SETS))])
batches = int(rng.integers(1, 4))
print("rows", t.nrows, "cond", cond, "keys", keys, "aggs", aggs, "opts", opts, "batches", batches)
for o in (opts, {"fast": 0, "spec": 0}):
    gpu, st = pu.run_gpu(t, cond, keys, aggs, batches=batches, **o)
    ora = n1o.run(t, cond, keys, aggs, threads=1)
    gm = {pu._canon_key(k): a for k, a in zip(gpu.keys, gpu.aggs)}
    om = {pu._canon_key(k): a for k, a in zip(ora.keys, ora.aggs)}
    print("opts", o, "stats", {k: st[k] for k in ("agg_mode", "spec_kernel", "rows_selected")}, "ora rows", ora.rows_passed)
    for k in sorted(set(gm) | set(om), key=str):
        if gm.get(k) != om.get(k):
            print("  DIFF", k, "gpu", gm.get(k), "ora", om.get(k))
