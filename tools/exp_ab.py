#!/usr/bin/env python3
"""Run the config-2 timing under several ablation builds of libn1k.so (N1K_LIB), interleaved rounds."""
import os, subprocess, sys, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [("base", "")] + [(os.path.basename(p)[10:-3], p) for p in sorted(glob.glob(os.path.join(ROOT, "tools/ab/*.so")))]
code = r'''
import sys, os
sys.path.insert(0, %r)
import torch, query_amd, bench
rows=100_000_000
D=bench.D
cols=bench.DeviceColumns(rows,1000,False,0,rows,0)
pj=query_amd.plan.filter_group_plan("(50 < %%s)" %% D("price"),[D("cat")],["sum(%%s)" %% D("price")])
op=query_amd.GpuFilterGroup(pj)
op.intern(bench.synth_dictionary(1000))
batch=[cols.by_path[p] for p in op.column_paths]
best=1e9
for _ in range(12):
    op.reopen(); op.process_device_items(rows,batch); r=op.after_items_raw(); best=min(best,op.stats()["device_ms"])
print("%%.4f" %% best)
''' % ROOT
for rnd in range(2):
    for name, path in libs:
        env = dict(os.environ)
        if path: env["N1K_LIB"] = path
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(rnd, name, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
