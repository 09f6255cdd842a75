#!/usr/bin/env python3
"""Ad-hoc scan-kernel experiments on the GPU box: prints the HIP-event time of the scan kernel per variant."""
import itertools
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import query_amd  # noqa
import bench  # noqa


def run(cols, rows, kcat, cond, keys, aggs, opts, reps=5, filter_only=False):
    pj = query_amd.plan.filter_group_plan(cond, keys, aggs, filter_only=filter_only)
    op = query_amd.GpuFilterGroup(pj)
    for k, v in opts.items():
        op.set_option(k, v)
    op.intern(bench.synth_dictionary(kcat))
    batch = [cols.by_path[p] for p in op.column_paths]
    best = 1e9
    wall = 1e9
    for _ in range(reps):
        op.reopen()
        t0 = time.perf_counter()
        op.process_device_items(rows, batch)
        r = op.after_items_raw()
        wall = min(wall, time.perf_counter() - t0)
        best = min(best, op.stats()["device_ms"])
    ng = r['ngroups']
    op.done()
    return best, wall * 1e3, ng


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    D = bench.D
    for kcat in (1000, 16):
        zipf = False
        cols = bench.DeviceColumns(rows, kcat, False, 0, rows, 0)
        c2 = ("(50 < %s)" % D("price"), [D("cat")], ["sum(%s)" % D("price")])
        allaggs = sorted(["count(*)", "sum(%s)" % D("price"), "avg(%s)" % D("price"), "min(%s)" % D("price"), "max(%s)" % D("price")])
        variants = []
        for slabs in (1, 0):
            for block, grid in ((1024, 256), (1024, 512), (512, 512), (512, 768), (512, 1024)):
                variants.append(("config2 spec b%d g%d slabs%d" % (block, grid, slabs), c2[0], c2[1], c2[2],
                                 {"block": block, "grid_blocks": grid, "slabs": slabs}, False))
        variants += [
            ("filter-only", c2[0], [], [], {}, True),
            ("le+avg,count jit", "(%s <= 50.5)" % D("price"), [D("cat")], ["avg(%s)" % D("price"), "count(*)"], {}, False),
            ("le+avg,count nojit", "(%s <= 50.5)" % D("price"), [D("cat")], ["avg(%s)" % D("price"), "count(*)"], {"jit": 0}, False),
            ("intkey sum spec", None, [D("region_id")], ["sum(%s)" % D("price")], {}, False),
            ("intkey sum interp", None, [D("region_id")], ["sum(%s)" % D("price")], {"fast": 0}, False),
            ("cat+region sum spec", None, [D("cat"), D("region_id")], ["sum(%s)" % D("price")], {}, False),
            ("cat+region sum interp", None, [D("cat"), D("region_id")], ["sum(%s)" % D("price")], {"fast": 0}, False),
            ("config2 fast(no spec)", c2[0], c2[1], c2[2], {"spec": 0}, False),
            ("config2 interp direct", c2[0], c2[1], c2[2], {"fast": 0}, False),
            ("nofilter sum", None, [D("cat")], ["sum(%s)" % D("price")], {}, False),
            ("nofilter count*", None, [D("cat")], ["count(*)"], {}, False),
            ("filter nokey count*", c2[0], [], ["count(*)"], {}, False),
            ("allaggs spec", c2[0], [D("cat")], allaggs, {}, False),
            ("allaggs spec noslab", c2[0], [D("cat")], allaggs, {"slabs": 0}, False),
        ]
        for name, cond, keys, aggs, opts, fo in variants:
            ms, wall, ng = run(cols, rows, kcat, cond, keys, aggs, opts, filter_only=fo)
            print("K=%-5d %-22s kernel %8.3f ms  wall %8.3f ms  groups %d  (%.0f GB/s @13B)" %
                  (kcat, name, ms, wall, ng, 13 * rows / ms / 1e6), flush=True)
        del cols
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
