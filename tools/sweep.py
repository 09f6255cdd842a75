#!/usr/bin/env python3
"""tools/sweep.py <workload> [--rows N] [--kcat K] "opt=val,opt=val" ... : ms per query of one bench.py workload under engine
option sets — one process, the synthetic columns generated once ("-" = no options)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--kcat", type=int, default=None)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("sets", nargs="+")
    a = ap.parse_args()
    if a.kcat is None:
        a.kcat = 100_000 if a.workload.startswith("config5") else 1000
    import torch
    import query_amd
    wl = bench.workloads()[a.workload]
    cols = bench.DeviceColumns(a.rows, a.kcat, False, 0, a.rows, 0)
    pj = query_amd.plan.filter_group_plan(wl["cond"], wl["keys"], wl["aggs"], order=wl.get("order"), limit=wl.get("limit"))
    for s in a.sets:
        op = query_amd.GpuFilterGroup(pj, device=0)
        for o in s.split(","):
            if o and o != "-":
                k, v = o.split("=")
                op.set_option(k, int(v))
        op.intern(bench.synth_dictionary(a.kcat))
        batch = op.make_device_batch(a.rows, [cols.by_path[p] for p in op.column_paths])

        def step():
            op.reopen()
            op.process_device_batch(batch)
            return op.after_items_raw()
        try:
            for _ in range(2):
                rows = step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                rows = step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.steps * 1e3
            st = op.stats()
            print("%-40s ms/query %.3f  device_ms %.3f  groups %d mode %s" % (s, ms, st["device_ms"], int(rows["ngroups"]), st["agg_mode"]), flush=True)
        except Exception as e:  # noqa: BLE001
            print("%-40s FAILED %s" % (s, e), flush=True)
        del op


if __name__ == "__main__":
    main()
