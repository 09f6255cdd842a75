#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X Filter -> Group -> Aggregate path.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic columns that are already resident in HBM:
reopen (drop the groups) -> scan kernel (Filter + InitialGroup + merge) -> FinalGroup -> groups copied to the host.
Workload at N=1 (BASELINE.json metric: rows/sec filter+group-by on 100M synthetic docs):
    config 2's query  SELECT cat, SUM(price) FROM default WHERE price > 50 GROUP BY cat   at 100 M rows, K_cat = 1000
(13 algorithmic bytes per row: price tag 1 + payload 8 + cat code 4).

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel (scan_spec_kernel + merge_slabs_kernel for
config 2) against the HBM peak from its HIP-event duration; `cpu_baseline` times the CPU oracle (a port of the reference's algorithm) on a
bounded sample of the same workload on this box's host cores.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
SEED = 0x5EED0001
METRIC = "rows/sec filter+group-by on 100M synthetic JSON docs; achieved HBM GB/s"  # BASELINE.json's metric, verbatim (the
# timed region runs over the documents' pre-extracted columns resident in HBM: config.workload says so; the rates from host
# buffers and from raw JSON text are the h2d_inclusive / json_end_to_end sub-records)
DTYPE = "int64/f64 tagged scalars (u8 tag + 8 B payload), u32 dictionary codes"


def D(*names):
    from query_amd import plan
    return plan.field_path("default", *names)


def workloads():
    return {
        "config2": {
            "sql": "SELECT cat, SUM(price) FROM default WHERE price > 50 GROUP BY cat",
            "cond": "(50 < %s)" % D("price"), "keys": [D("cat")], "aggs": ["sum(%s)" % D("price")],
            "bytes_per_row": 13,
        },
        "config2_allaggs": {
            "sql": "SELECT cat, COUNT(*), SUM(price), AVG(price), MIN(price), MAX(price) FROM default WHERE price > 50 GROUP BY cat",
            "cond": "(50 < %s)" % D("price"), "keys": [D("cat")],
            "aggs": sorted(["count(*)", "sum(%s)" % D("price"), "avg(%s)" % D("price"), "min(%s)" % D("price"),
                            "max(%s)" % D("price")]),
            "bytes_per_row": 13,
        },
        "config3": {
            "sql": "SELECT cat, COUNT(DISTINCT user_id), AVG(price) FROM default GROUP BY cat",
            "cond": None, "keys": [D("cat")],
            "aggs": sorted(["count(distinct %s)" % D("user_id"), "avg(%s)" % D("price")]),
            "bytes_per_row": 22,
        },
        "config5": {
            "sql": "SELECT cat, region_id, SUM(price) s FROM default GROUP BY cat, region_id ORDER BY s DESC LIMIT 100",
            "cond": None, "keys": [D("cat"), D("region_id")], "aggs": ["sum(%s)" % D("price")],
            "order": [("sum(%s)" % D("price"), True)], "limit": 100,
            "bytes_per_row": 22,
        },
        # arithmetic in the Filter and in the aggregate's operand (tests' ARITH_CASES[0]) next to the same shape without it:
        # the run-time-built scan evaluates the nodes in registers, both read the same 22 B per row
        "arith": {
            "sql": "SELECT cat, COUNT(*), SUM(price * region_id) FROM default WHERE price + region_id > 100 GROUP BY cat",
            "cond": "(100 < (%s + %s))" % (D("price"), D("region_id")), "keys": [D("cat")],
            "aggs": ["count(*)", "sum((%s * %s))" % (D("price"), D("region_id"))],
            "bytes_per_row": 22,
        },
        "arith_filter": {   # (only the Filter's node)
            "sql": "SELECT cat, COUNT(*), SUM(region_id) FROM default WHERE price + region_id > 100 GROUP BY cat",
            "cond": "(100 < (%s + %s))" % (D("price"), D("region_id")), "keys": [D("cat")],
            "aggs": ["count(*)", "sum(%s)" % D("region_id")], "bytes_per_row": 22,
        },
        "arith_sum": {      # (only the aggregate's node)
            "sql": "SELECT cat, COUNT(*), SUM(price * region_id) FROM default WHERE price > 50 GROUP BY cat",
            "cond": "(50 < %s)" % D("price"), "keys": [D("cat")],
            "aggs": ["count(*)", "sum((%s * %s))" % (D("price"), D("region_id"))], "bytes_per_row": 22,
        },
        "arith_plain": {
            "sql": "SELECT cat, COUNT(*), SUM(region_id) FROM default WHERE price > 50 GROUP BY cat",
            "cond": "(50 < %s)" % D("price"), "keys": [D("cat")],
            "aggs": ["count(*)", "sum(%s)" % D("region_id")],
            "bytes_per_row": 22,
        },
        # Filter alone (execution/filter.go:49-61): ascending ordinals of the rows whose condition is TRUE.  9 B per row read,
        # 8 B per survivor written (half the rows pass: 13 B per row in all)
        "filter": {
            "sql": "SELECT RAW meta().id FROM default WHERE price > 50   (Filter only: the survivors' row ordinals)",
            "cond": "(50 < %s)" % D("price"), "keys": [], "aggs": [], "filter_only": True, "bytes_per_row": 13,
        },
        "config5_keys": {
            "sql": "SELECT cat, region_id, SUM(price) FROM default GROUP BY cat, region_id",
            "cond": None, "keys": [D("cat"), D("region_id")], "aggs": ["sum(%s)" % D("price")],
            "bytes_per_row": 22,
        },
    }


class DeviceColumns:
    """Synthetic columns generated on the device (n1k_synth_columns), held in torch tensors."""

    def __init__(self, nrows: int, k_cat: int, zipf: bool, first_row: int, total_rows: int, device: int):
        import torch
        from query_amd import _ffi
        dev = torch.device("cuda", device)
        self.nrows = nrows
        self.cat = torch.empty(nrows, dtype=torch.int32, device=dev)
        self.price_t = torch.empty(nrows, dtype=torch.uint8, device=dev)
        self.price_p = torch.empty(nrows, dtype=torch.int64, device=dev)
        self.user_t = torch.empty(nrows, dtype=torch.uint8, device=dev)
        self.user_p = torch.empty(nrows, dtype=torch.int64, device=dev)
        self.region_t = torch.empty(nrows, dtype=torch.uint8, device=dev)
        self.region_p = torch.empty(nrows, dtype=torch.int64, device=dev)
        cdf = torch.from_numpy(zipf_cdf(k_cat)).to(dev) if zipf else None
        spec = _ffi.SynthSpec(SEED, first_row, nrows, total_rows, k_cat, 1 if zipf else 0,
                              cdf.data_ptr() if zipf else None)
        st = _ffi.lib().n1k_synth_columns(device, None, C.byref(spec), self.cat.data_ptr(), self.price_t.data_ptr(),
                                          self.price_p.data_ptr(), self.user_t.data_ptr(), self.user_p.data_ptr(),
                                          self.region_t.data_ptr(), self.region_p.data_ptr())
        if st != 0:
            raise RuntimeError("n1k_synth_columns failed: %d" % st)
        torch.cuda.synchronize(dev)
        from query_amd import _ffi as f
        self.by_path = {
            D("cat"): (f.COL_DICT32, None, None, self.cat.data_ptr()),
            D("price"): (f.COL_TAGGED64, self.price_t.data_ptr(), self.price_p.data_ptr(), None),
            D("user_id"): (f.COL_TAGGED64, self.user_t.data_ptr(), self.user_p.data_ptr(), None),
            D("region_id"): (f.COL_TAGGED64, self.region_t.data_ptr(), self.region_p.data_ptr(), None),
        }


def zipf_cdf(k: int):
    """Zipf(s=1) cdf over k categories, same sequential float64 arithmetic as the C generators."""
    import numpy as np
    h = 0.0
    for i in range(k):
        h += 1.0 / float(i + 1)
    acc, out = 0.0, np.zeros(k, dtype=np.float64)
    for i in range(k):
        acc += (1.0 / float(i + 1)) / h
        out[i] = acc
    if k:
        out[k - 1] = 1.0
    return out


def synth_dictionary(k_cat: int):
    return [b"cat_%d" % i for i in range(k_cat)] + [b"n/a"]


def cpu_baseline(wl: dict, k_cat: int, zipf: bool, total_rows: int, sample_rows: int) -> dict:
    """Time the CPU oracle (port of the reference algorithm: Parallel copies with private maps + serial merge,
    execution/parallel.go:52-75) on the first rows of the same data set, at T = 1, T = 16 (the GPU box's CPU share for
    one GPU) and T = every core this process may run on.  Every leg is bounded to a few seconds: the sample shrinks for
    T = 1.  `value` / `cores` are the fastest leg's; `by_threads` holds them all."""
    from oracle import n1o
    ncores = len(os.sched_getaffinity(0))
    legs = sorted({1, min(16, ncores), ncores})
    t = n1o.synth_table(sample_rows, k_cat=k_cat, zipf=zipf, seed=SEED, first_row=0, total_rows=total_rows)
    by, best = {}, None
    rate16 = None
    for T in sorted(legs, reverse=True):  # many threads first: their rate bounds the single-thread sample
        n = sample_rows
        if T == 1 and rate16:
            n = int(max(100_000, min(sample_rows, rate16 / 8 * 4.0)))  # ~4 s if one thread does 1/8 of the 16-thread rate
        tt = t if n == sample_rows else n1o.Table([n1o.Column(c.name, c.kind, tags=None if c.tags is None else c.tags[:n],
                                                              payload=None if c.payload is None else c.payload[:n],
                                                              codes=None if c.codes is None else c.codes[:n]) for c in t.columns],
                                                  t.dictionary)
        res = n1o.run(tt, wl["cond"], wl["keys"], wl["aggs"], threads=T)
        rate = n / res.seconds
        by[str(T)] = {"value": rate, "rows": n, "seconds": res.seconds}
        if T == min(16, ncores):
            rate16 = rate
        if best is None or rate > best[0]:
            best = (rate, T, n, res.seconds)
    return {"value": best[0], "unit": "rows/s", "cores": best[1], "kind": "port",
            "sample": "first %d rows of the same synthetic data set, %d threads, %.2f s wall (fastest of T = %s; host has %d cores)" %
                      (best[2], best[1], best[3], "/".join(str(x) for x in legs), ncores),
            "by_threads": by}


def ingest_rates(args, wl, cols) -> dict:
    """the same query fed from HOST column buffers (n1k_push_batch: PCIe included) and from
    raw JSON documents (n1k_push_json: host extraction + H2D + kernels).  Never the headline `value`."""
    import numpy as np
    import query_amd
    from query_amd import _ffi, plan
    out = {}
    rows = min(args.rows, args.ingest_rows)
    batch_rows = 4_000_000

    class Col:
        def __init__(self, kind, tags=None, payload=None, codes=None):
            self.kind, self.tags, self.payload, self.codes = kind, tags, payload, codes

    pj = plan.filter_group_plan(wl["cond"], wl["keys"], wl["aggs"], order=wl.get("order"), limit=wl.get("limit"))
    op = query_amd.GpuFilterGroup(pj)
    op.intern(synth_dictionary(args.kcat))
    host = {D("cat"): Col(_ffi.COL_DICT32, codes=cols.cat[:rows].cpu().numpy().view(np.uint32)),
            D("price"): Col(_ffi.COL_TAGGED64, tags=cols.price_t[:rows].cpu().numpy(), payload=cols.price_p[:rows].cpu().numpy().view(np.uint64)),
            D("user_id"): Col(_ffi.COL_TAGGED64, tags=cols.user_t[:rows].cpu().numpy(), payload=cols.user_p[:rows].cpu().numpy().view(np.uint64)),
            D("region_id"): Col(_ffi.COL_TAGGED64, tags=cols.region_t[:rows].cpu().numpy(), payload=cols.region_p[:rows].cpu().numpy().view(np.uint64))}
    use = [host[p] for p in op.column_paths]

    def sl(c, lo, hi):
        return Col(c.kind, None if c.tags is None else c.tags[lo:hi], None if c.payload is None else c.payload[lo:hi],
                   None if c.codes is None else c.codes[lo:hi])

    best = None
    for _ in range(3):
        op.reopen()
        t0 = time.perf_counter()
        for lo in range(0, rows, batch_rows):
            op.process_items([sl(c, lo, min(rows, lo + batch_rows)) for c in use], remap=False)
        op.after_items_raw()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out["h2d_inclusive"] = {"value": rows / best, "unit": "rows/s", "rows": rows, "batch_rows": batch_rows,
                            "GB/s": rows * wl["bytes_per_row"] / best / 1e9,
                            "what": "n1k_push_batch from pageable host columns (H2D over PCIe + kernels) + n1k_finish, best of 3"}
    op.done()

    # raw documents of the synthetic data set (id, cat, price, user_id, region_id + padding, ~150 B each), written by
    # the library's own formatter from the host copies of the columns (n1k_synth_documents: 10 M documents in about a second)
    ndocs = min(rows, args.json_docs)
    lib = _ffi.lib()
    blob = np.empty(ndocs * (175 + 64), dtype=np.uint8)
    offsets = np.empty(ndocs + 1, dtype=np.uint64)
    used = C.c_size_t(0)
    st = lib.n1k_synth_documents(ndocs, 0, host[D("cat")].codes.ctypes.data, host[D("price")].tags.ctypes.data,
                                 host[D("price")].payload.ctypes.data, host[D("user_id")].payload.ctypes.data,
                                 host[D("region_id")].payload.ctypes.data, 64, blob.ctypes.data, blob.size, offsets.ctypes.data, C.byref(used))
    if st != 0:
        raise RuntimeError("n1k_synth_documents failed: %d" % st)
    nbytes = int(used.value)
    op = query_amd.GpuFilterGroup(pj)
    optr = offsets.ctypes.data_as(C.POINTER(C.c_uint64))
    bptr = C.cast(blob.ctypes.data, C.c_char_p)
    best = None
    for _ in range(3):
        op.reopen()
        t0 = time.perf_counter()
        op._check(op._lib.n1k_push_json(op._h, ndocs, optr, bptr))
        res = op.after_items_raw()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    jst = op.stats()
    assert int(res["ngroups"]) == args.kcat, "json_end_to_end: %d groups" % int(res["ngroups"])
    out["json_end_to_end"] = {"value": ndocs / best, "unit": "docs/s", "docs": ndocs, "json_bytes": nbytes,
                              "GB/s_of_text": nbytes / best / 1e9, "docs_extracted_on_the_device": int(jst["json_device_docs"]),
                              "what": "n1k_push_json (document bytes H2D -> json_extract_kernel -> columns in HBM -> scan kernels; the batch's "
                                      "distinct strings and the documents the kernel leaves alone visit the host) + n1k_finish, best of 3"}
    op.done()
    return out


def kernel_label(st: dict, wl: dict) -> str:
    """What the batches' kernels are for this workload (`kernel_split["batch kernels"]`: the HIP-event time of n1k_push_device_batch's launches)."""
    if st.get("agg_mode") == 4:
        return "key probe + scan_spec_records_kernel + radix_scatter_sub_kernel + agg_bins16_kernel"
    if any("distinct" in a for a in wl["aggs"]):
        return "scan_spec_kernel incl. the member words' first partition pass (+merge_slabs_kernel)"
    if st.get("spec_kernel") == 3:
        return "run-time-built scan_spec_body with the plan's arithmetic in registers (+merge_slabs_kernel)"
    return "scan_spec_kernel(+merge_slabs_kernel)" if st.get("spec_kernel") else "scan_fast/scan_group_kernel"


def roofline_of(wl: dict, rows: int, st: dict, workload: str, kcat: int, opts) -> dict:
    """ONE definition for every workload: the query's algorithmic bytes (rows x the referenced columns' widths)
    over the HIP-event time of the WHOLE query on the handle's stream — reopen, every kernel of the batches, DISTINCT sets /
    partition passes / top-k, FinalGroup, and the gaps between them (n1k_stats.query_ms).  The split by kernel family is the
    sub-record `kernel_split`; `traffic` is the whole query's HBM traffic from the committed PMC passes of the same command."""
    alg = wl["bytes_per_row"] * rows
    if wl.get("filter_only"):
        # the Filter-only kernel alone: its survivors' ordinals (8 B each) then cross PCIe to the host, which is the consumer's
        # cost at the link's 52 GB/s, not the kernel's
        alg = 9 * rows + 8 * int(st["rows_selected"])
        ach = alg / (st["device_ms"] * 1e-3) / 1e9 if st["device_ms"] else 0.0
        return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                "what": "Filter-only kernel (predicate + ordered compaction in one pass): 9 B per row read + 8 B per survivor written / its "
                        "HIP-event time; the copy of the ordinals to the host is not in it",
                "query_ms": st["device_ms"], "algorithmic_bytes_per_launch": alg}
    q = st.get("query_ms") or 0.0
    ach = alg / (q * 1e-3) / 1e9 if q > 0 else 0.0
    r = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
         "what": "whole query: algorithmic bytes / HIP-event time from reopen to FinalGroup's last kernel on the handle's stream",
         "query_ms": q, "algorithmic_bytes_per_launch": alg,
         "kernel_split": {"batch kernels": {"ms": st["device_ms"], "what": kernel_label(st, wl),
                                            "achieved_GB/s": alg / (st["device_ms"] * 1e-3) / 1e9 if st["device_ms"] else None},
                          "rest of the query (reopen, sets / passes of n1k_finish, FinalGroup, top-k, gaps)": {"ms": max(0.0, q - st["device_ms"])}}}
    # HBM traffic of the whole query from the committed PMC passes of this same command (rocprofv3 --pmc cannot run inside
    # the timed process): per kernel 2 x FETCH_SIZE (MI355X_MICROARCH.md, HBM: gfx950 counts a wide coalesced read at half its
    # bytes) + WRITE_SIZE, KB per dispatch x dispatches per query, summed over the query's kernels.  The figure belongs to
    # the build the profile was taken from (`source_hash`; tools/profile_round.sh): re-profile after kernel changes.
    pmc = None
    for tag in ("r03", "r02"):
        f = os.path.join(ROOT, "profiles", "%s_bench_%s_100M_pmc_fetch_write.json" % (tag, workload))
        if os.path.exists(f):
            pmc = f
            break
    if pmc and rows == 100_000_000 and kcat == (100_000 if workload.startswith("config5") else 1000) and not opts:
        try:
            with open(pmc) as fh:
                c = json.load(fh)
            from query_amd import build as qbuild
            if c.get("source_hash") in (None, qbuild.source_hash()):  # (None: profiles taken before the stamp existed)
                nq = c.get("queries") or min(v["dispatches"] for k, v in c["FETCH_SIZE"].items() if "scan_spec" in k or "n1k_jit" in k)
                tot = 0.0
                for k, v in c["FETCH_SIZE"].items():
                    # (not the query's: the data generator, the Filter-only count bench.py checks the survivors against and the
                    #  copies of that check's ordinals to the host — a query's own copies are counters and a few rows: KBs)
                    if "synth_kernel" in k or ("filter_" in k and workload != "filter"):
                        continue
                    if "copyBuffer" in k and workload != "filter" and v["avg_KB"] > 1024.0:
                        continue
                    w = c["WRITE_SIZE"].get(k, {"avg_KB": 0.0})
                    tot += 1024.0 * (2.0 * v["avg_KB"] + w["avg_KB"]) * v["dispatches"] / nq
                r["traffic"] = tot
                r["traffic_source"] = "profiles/%s (sum over the query's kernels of 2*FETCH_SIZE + WRITE_SIZE)" % os.path.basename(pmc)
            else:
                r["traffic_source"] = "none: profiles/%s belongs to other kernel sources (%s)" % (os.path.basename(pmc), c.get("source_hash"))
        except Exception as e:
            r["traffic_source"] = "none: %r" % (e,)
    return r


def run_resident(workload: str, rows: int, kcat: int, zipf: bool, local_rank: int, steps: int, warmup: int, opts=(), cols=None,
                 three_calls=False) -> dict:
    """`steps` timed executions of one workload over columns resident in HBM (n1k_run_device_batch: reopen + scan + FinalGroup in
    one call through the ABI).  Checks once, outside the timed region, that the groups are all there and that the Filter's
    survivors equal an independent Filter-only count."""
    import torch
    import query_amd
    wl = workloads()[workload]
    own = cols is None
    if own:
        cols = DeviceColumns(rows, kcat, zipf, 0, rows, local_rank)
    pj = query_amd.plan.filter_group_plan(wl["cond"], wl["keys"], wl["aggs"], order=wl.get("order"), limit=wl.get("limit"),
                                          filter_only=bool(wl.get("filter_only")))
    op = query_amd.GpuFilterGroup(pj, device=local_rank)
    for o in opts:
        k, v = o.split("=")
        op.set_option(k, int(v))
    op.intern(synth_dictionary(kcat))
    batch = op.make_device_batch(rows, [cols.by_path[p] for p in op.column_paths])

    def step():
        if three_calls:
            op.reopen()
            op.process_device_batch(batch)
            return op.after_items_raw()
        return op.run_device_batch_raw(batch)

    for _ in range(warmup):
        res = step()
    op.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = step()
    op.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    st = op.stats()  # (reopen zeroes the stats each step: these are the last query's)
    ngroups = int(res["ngroups"])
    if wl.get("filter_only"):
        sel = res["selected"]
        assert len(sel) == st["rows_selected"] and (len(sel) < 2 or bool((sel[1:] > sel[:-1]).all())), "ordinals not ascending"
    # parity seams, outside the timed region: every group of the key domain is there (uniform keys at these sizes), and an
    # independent kernel family (Filter-only: mask + compaction) counts the same survivors
    checks = {}
    if len(wl["keys"]) == 1 and not wl.get("limit") and not zipf and rows >= 1000 * kcat:
        assert ngroups == kcat, "%s: %d groups, the key domain has %d" % (workload, ngroups, kcat)
        checks["groups == K_cat"] = True
    if wl["cond"] and not wl.get("filter_only"):
        fo = query_amd.GpuFilterGroup(query_amd.plan.filter_group_plan(wl["cond"], [], [], filter_only=True), device=local_rank)
        fo.intern(synth_dictionary(kcat))
        fb = fo.make_device_batch(rows, [cols.by_path[p] for p in fo.column_paths])
        fo.reopen()
        fo.process_device_batch(fb)
        nsel = len(fo.after_items_raw()["selected"])
        fo.done()
        assert nsel == st["rows_selected"], "%s: the scan kept %d rows, the Filter-only kernels %d" % (workload, st["rows_selected"], nsel)
        checks["rows_selected == Filter-only count"] = True
    out = {"workload": workload, "sql": wl["sql"], "rows": rows, "k_cat": kcat, "zipf": bool(zipf), "steps": steps,
           "ms_per_step": elapsed / steps * 1e3, "value": rows * steps / elapsed, "unit": "rows/s", "groups": ngroups,
           "rows_selected": st["rows_selected"], "agg_mode": st["agg_mode"], "spec_kernel": st["spec_kernel"],
           "roofline": roofline_of(wl, rows, st, workload, kcat, opts), "checks": checks}
    op.done()
    if own:
        del cols, batch
        torch.cuda.empty_cache()
    return out


def by_rows(args, local_rank: int) -> dict:
    """north_star: rows/sec at 10 M / 100 M / 1 B rows.  The same query at the two other sizes on this GPU (columns resident
    in HBM, 1 B rows = 13 GB of config 2's columns): 10 untimed + 10 timed executions each.  Never the headline `value`."""
    import torch
    out = {}
    for label, n in (("10M", 10_000_000), ("1B", 1_000_000_000)):
        if n == args.rows:
            continue
        free, _total = torch.cuda.mem_get_info(local_rank)
        if n * 31 * 1.2 > free:  # (the generator fills all seven arrays: 31 B per row)
            out[label] = {"skipped": "not enough free HBM"}
            continue
        r = run_resident("config2", n, args.kcat, False, local_rank, 10, 10)
        out[label] = {"rows": n, "ms_per_step": r["ms_per_step"], "value": r["value"], "unit": "rows/s", "query_ms": r["roofline"]["query_ms"],
                      "frac": r["roofline"]["frac"], "batch_kernels_ms": r["roofline"]["kernel_split"]["batch kernels"]["ms"]}
    return out


def by_config(args, local_rank: int, cols) -> dict:
    """BASELINE.json's other single-GPU query shapes at 100 M rows in the driver's own record: config 3 (COUNT(DISTINCT user_id)
    + AVG(price) GROUP BY cat — the only 100 M-row single-GPU entry of `configs`) over the headline's resident columns, and
    one shard of config 5 (GROUP BY cat, region_id ORDER BY SUM(price) DESC LIMIT 100, K_cat = 100 000: 6.4 M groups).  Whole
    query wall time per step, whole-query roofline fraction, kernel split.  Never the headline `value`."""
    out = {}
    out["config3"] = run_resident("config3", args.rows, 1000, False, local_rank, 10, 3, cols=cols if args.kcat == 1000 else None)
    out["config5"] = run_resident("config5", args.rows, 100_000, False, local_rank, 10, 3)
    return out


def zipf_record(args, local_rank: int) -> dict:
    """Skew (cat ~ Zipf(s = 1.0)): config 2 at 100 M rows with Zipf keys over K = 1000 and K = 16 categories next to
    uniform keys over K = 16 (K = 1000 uniform is the headline)."""
    out = {}
    for label, k, z in (("zipf_K1000", 1000, True), ("uniform_K16", 16, False), ("zipf_K16", 16, True)):
        r = run_resident("config2", args.rows, k, z, local_rank, 10, 3)
        out[label] = {"ms_per_step": r["ms_per_step"], "value": r["value"], "unit": "rows/s", "query_ms": r["roofline"]["query_ms"],
                      "frac": r["roofline"]["frac"], "batch_kernels_ms": r["roofline"]["kernel_split"]["batch kernels"]["ms"], "groups": r["groups"]}
    return out


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher's environment: start the N ranks as child processes (one per GPU, the
    same command line, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), relay rank 0's JSON line, fail if any rank fails.
    Runs BEFORE this process has imported torch or touched HIP: a process that has initialised the GPU must not start others."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = b""
    deadline = None
    rcs = [None] * len(procs)
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                try:
                    if i == 0:
                        o, _ = p.communicate(timeout=0.2)
                        out0 += o or b""
                    else:
                        p.wait(timeout=0.2)
                    rcs[i] = p.returncode
                except subprocess.TimeoutExpired:
                    pass
        if deadline is None and any(rc not in (None, 0) for rc in rcs):
            deadline = time.time() + 30  # a rank died: its peers get half a minute to notice, then they are stopped (by PID)
        if deadline is not None and time.time() > deadline:
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    p.kill()
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    bad = [(i, rc) for i, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write("bench.py: ranks failed (rank, exit code): %r\n" % bad)
        return 1
    return 0


def dry_run(args, rank: int, world: int):
    """--dry-run: the N > 1 start-up without a GPU — rendezvous over gloo, the communicator id from rank 0 to every rank, the
    shard bounds of the synthetic data set, one all-reduce; rank 0 prints a JSON line.  (tests/test_distributed_cpu.py)"""
    import numpy as np
    import torch
    import torch.distributed as dist
    from query_amd import distributed as qd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ident = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        ident = torch.from_numpy(np.frombuffer(os.urandom(128), dtype=np.uint8).copy())
    dist.broadcast(ident, src=0)
    total_rows, first, rows = qd.shard_bounds(args, rank, world)
    check = torch.tensor([rows, int(ident.to(torch.int64).sum())], dtype=torch.int64)
    gathered = [torch.zeros_like(check) for _ in range(world)]
    dist.all_gather(gathered, check)
    ok = sum(int(g[0]) for g in gathered) == total_rows and len({int(g[1]) for g in gathered}) == 1
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "total_rows": total_rows, "rows_per_rank": [int(g[0]) for g in gathered],
                          "scaling": "strong" if args.total_rows else "weak", "ranks_agree": ok}))
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=100_000_000, help="rows per GPU (weak scaling)")
    ap.add_argument("--total-rows", type=int, default=0, help="multi-GPU: rows in all, split over the ranks (strong scaling: "
                    "BASELINE config 4 = 100 M, config 5 = 1 B over 8 GPUs); 0 = --rows per GPU")
    ap.add_argument("--workload", default="config2")
    ap.add_argument("--kcat", type=int, default=None, help="distinct cat values (default: 1000; 100000 for config5*, as BASELINE.json names them)")
    ap.add_argument("--zipf", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=20_000_000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--three-calls", action="store_true", help="n1k_reset / n1k_push_device_batch / n1k_finish as three calls from Python")
    ap.add_argument("--no-sizes", action="store_true", help="skip the by_rows sub-record (the same query at 10 M and 1 B rows)")
    ap.add_argument("--no-ingest", action="store_true", help="skip the h2d_inclusive / json_end_to_end sub-records")
    ap.add_argument("--ingest-rows", type=int, default=40_000_000, help="rows pushed from host buffers for h2d_inclusive")
    ap.add_argument("--json-docs", type=int, default=10_000_000, help="documents pushed as raw JSON for json_end_to_end")
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value")
    ap.add_argument("--exchange", default="auto", choices=["auto", "rows", "partials", "gathered"],
                    help="multi-GPU: what crosses xGMI.  auto = rows (the configuration north_star names: filtered rows "
                         "hash-partitioned on the group key by one RCCL all-to-all); partials = per-GPU partial groups "
                         "hash-partitioned to owners; gathered = partial groups all-gathered while they are few (the "
                         "ablation: G groups travel instead of the rows); plans with DISTINCT always exchange rows")
    ap.add_argument("--no-ablation", action="store_true", help="multi-GPU: skip the partial-group ablation measured next to the row exchange")
    ap.add_argument("--force-dist", action="store_true", help="take the multi-rank code path even with one rank")
    ap.add_argument("--dry-run", action="store_true", help="N > 1 start-up only, over gloo, no GPU (launcher, rendezvous, shard bounds)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # no launcher: be one (before anything touches torch / HIP)
        raise SystemExit(launch_ranks(args))
    if args.kcat is None:
        args.kcat = 100_000 if args.workload.startswith("config5") else 1000

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, rank, world)

    import torch
    import query_amd
    if query_amd.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the device path has no CPU fallback")
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if query_amd.device_count() < local_world:  # (every rank of the node sees the same count: all of them stop here, none waits)
        raise SystemExit("bench.py --gpus %d: %d ranks on this node but %d visible GPU(s); one rank per GPU" %
                         (args.gpus, local_world, query_amd.device_count()))
    torch.cuda.set_device(local_rank)

    if world > 1 or args.force_dist:
        from query_amd import distributed as qd
        return qd.bench_main(args, rank, world, local_rank)

    wl = workloads()[args.workload]
    total_rows = args.rows
    cols = DeviceColumns(args.rows, args.kcat, bool(args.zipf), 0, total_rows, local_rank)
    r = run_resident(args.workload, args.rows, args.kcat, bool(args.zipf), local_rank, args.steps, args.warmup, opts=args.opt, cols=cols,
                     three_calls=args.three_calls)
    out = {
        "metric": METRIC,
        "value": r["value"],
        "unit": "rows/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": r["ms_per_step"],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": DTYPE,
        "data": "synthetic",
        "config": {"workload": "%s: %s @ %d rows, K_cat=%d%s, columns resident in HBM" %
                               (args.workload, wl["sql"], args.rows, args.kcat, " zipf" if args.zipf else ""),
                   "rows_per_gpu": args.rows, "groups": r["groups"], "rows_selected": r["rows_selected"],
                   "agg_mode": r["agg_mode"], "spec_kernel": r["spec_kernel"], "checks": r["checks"]},
        "roofline": r["roofline"],
    }
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(wl, args.kcat, bool(args.zipf), total_rows,
                                           min(args.cpu_sample, args.rows))
    if not args.no_ingest and not args.zipf:
        try:
            out.update(ingest_rates(args, wl, cols))
        except Exception as e:  # the sub-records never cost the headline line
            out["ingest_error"] = repr(e)[:300]
    if not args.no_sizes and args.workload == "config2" and not args.zipf and not args.opt:
        keep = [cols]
        for name, fn in (("by_config", lambda: by_config(args, local_rank, keep[0])), ("zipf", lambda: zipf_record(args, local_rank)),
                         ("by_rows", lambda: by_rows(args, local_rank))):
            if name == "by_rows":
                cols = keep[0] = None  # (1 B rows need the room)
                torch.cuda.empty_cache()
            try:
                out[name] = fn()
            except Exception as e:  # the sub-records never cost the headline line
                out[name] = {"error": repr(e)[:300]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
