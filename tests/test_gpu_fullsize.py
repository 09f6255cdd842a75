"""Parity at BASELINE.json's full size (100 M rows, generated on the device) through size-independent properties:
the oracle cannot run 100 M rows in seconds, so the HIP path is checked against itself across independent kernels
and batchings — a checksum of counts, linearity of SUM/COUNT over a split of the input, idempotence, and the same
answer from the specialised and the interpreted kernels — plus the oracle on a prefix of the same data set."""
import ctypes as C

import numpy as np
import pytest

import parity_util as pu
import query_amd
from oracle import n1o
from query_amd import _ffi, plan

pytestmark = pytest.mark.gpu

ROWS = 100_000_000
K_CAT = 1000


def D(*names):
    return plan.field_path("default", *names)


COND = "(50 < %s)" % D("price")
KEYS = [D("cat")]
AGGS = sorted(["count(*)", "sum(%s)" % D("price"), "max(%s)" % D("price"), "min(%s)" % D("user_id")])


@pytest.fixture(scope="module")
def columns():
    import bench
    return bench.DeviceColumns(ROWS, K_CAT, False, 0, ROWS, 0)


def run(columns, cond, keys, aggs, ranges, filter_only=False, kcat=K_CAT, order=None, limit=None, **opts):
    """Push the given row ranges of the resident columns as separate device batches."""
    import bench
    op = query_amd.GpuFilterGroup(plan.filter_group_plan(cond, keys, aggs, filter_only=filter_only, order=order, limit=limit), **opts)
    op.intern(bench.synth_dictionary(kcat))
    for lo, hi in ranges:
        cols = []
        for p in op.column_paths:
            kind, tags, pay, codes = columns.by_path[p]
            if kind == _ffi.COL_DICT32:
                cols.append((kind, None, None, codes + 4 * lo))
            else:
                cols.append((kind, tags + lo, pay + 8 * lo, None))
        op.process_device_items(hi - lo, cols)
    raw = op.after_items_raw()
    stats = op.stats()
    op.done()
    return raw, stats


def as_dict(raw):
    """code of the cat key -> tuple of (tag, bits) per aggregate"""
    out = {}
    for g in range(raw["ngroups"]):
        out[int(raw["keys"][g, 0]["v"])] = tuple((int(a["tag"]), int(a["v"])) for a in raw["aggs"][g])
    return out


def close(a, b, rel=1e-9):
    if a[0] != b[0]:
        return False
    if a[0] == _ffi.T_FLOAT:
        x, y = np.uint64(a[1]).view(np.float64), np.uint64(b[1]).view(np.float64)
        return x == y or abs(x - y) <= rel * max(abs(x), abs(y))
    return a[1] == b[1]


def test_full_size_properties(columns):
    whole, st = run(columns, COND, KEYS, AGGS, [(0, ROWS)])
    # 13 accumulator words x 1002 slots = 104 KB of LDS table: beyond what the plan-specialised kernels take (64 KB), so
    # the bounded-shape kernel ran, with the perfect-hash table
    assert st["spec_kernel"] == 0 and st["agg_mode"] == _ffi.MODE_LDS_DIRECT
    d_whole = as_dict(whole)
    assert whole["ngroups"] == K_CAT
    ci = AGGS.index("count(*)")
    # checksum of checksums: the per-group COUNT(*) add up to the rows the Filter kept, which an independent kernel
    # family (Filter-only: ballot mask + compaction) counts again
    total = sum(v[ci][1] for v in d_whole.values())
    assert total == st["rows_selected"]
    sel, fst = run(columns, COND, [], [], [(0, ROWS)], filter_only=True)
    assert fst["rows_selected"] == total == len(sel["selected"])
    s = sel["selected"]
    assert np.all(s[1:] > s[:-1])  # ascending row ordinals
    # linearity / batching: 4 unequal batches (odd boundaries -> unaligned 16-byte loads fall back) == one batch
    cuts = [0, 24_999_999, 50_000_001, 75_000_003, ROWS]
    parts, _ = run(columns, COND, KEYS, AGGS, list(zip(cuts[:-1], cuts[1:])))
    d_parts = as_dict(parts)
    assert d_parts.keys() == d_whole.keys()
    for k in d_whole:
        for a, b in zip(d_whole[k], d_parts[k]):
            assert close(a, b), (k, a, b)
    # the interpreted kernel (open-addressed LDS hash) is an independent implementation of the same semantics
    interp, ist = run(columns, COND, KEYS, AGGS, [(0, ROWS)], fast=0, agg_mode=_ffi.MODE_LDS_HASH)
    d_interp = as_dict(interp)
    for k in d_whole:
        for a, b in zip(d_whole[k], d_interp[k]):
            assert close(a, b), (k, a, b)
    # idempotence: the same query twice
    again, _ = run(columns, COND, KEYS, AGGS, [(0, ROWS)])
    for k, v in as_dict(again).items():
        for a, b in zip(v, d_whole[k]):
            assert close(a, b)


def test_headline_kernel_at_its_benchmarked_size(columns):
    """bench.py's headline is config 2's own plan (SELECT cat, SUM(price) ... WHERE price > 50 GROUP BY cat) at 100 M rows on
    the PREBUILT plan-specialised kernel scan_spec_kernel<Spec_gt_sum, 2, 512, true> (stats.spec_kernel == 1).  That very
    instantiation, at that size, through the one-call entry point the bench times (n1k_run_device_batch): every group present,
    the Filter's survivor count equal to an independent Filter-only count, batching invariance (the same kernel over unequal
    batches, and twice in a row: the one-call path leaves the device clean behind it), the same groups as the kernel family
    test_full_size_properties pins (per-group SUM from the 4-aggregate plan on the bounded-shape kernel), and the oracle on a
    prefix."""
    import bench
    aggs = ["sum(%s)" % D("price")]
    op = query_amd.GpuFilterGroup(plan.filter_group_plan(COND, KEYS, aggs))
    op.intern(bench.synth_dictionary(K_CAT))
    batch = op.make_device_batch(ROWS, [columns.by_path[p] for p in op.column_paths])
    first = op.run_device_batch_raw(batch)
    st = op.stats()
    assert st["spec_kernel"] == 1 and st["agg_mode"] == _ffi.MODE_LDS_DIRECT  # the prebuilt instantiation, perfect-hash table
    second = op.run_device_batch_raw(batch)  # (no reopen kernel in front of this one: the first left the device clean)
    third = op.run_device_batch_raw(batch)
    op.done()
    d1, d2, d3 = as_dict(first), as_dict(second), as_dict(third)
    assert first["ngroups"] == K_CAT and d1.keys() == d2.keys() == d3.keys()
    for k in d1:
        assert close(d1[k][0], d2[k][0]) and close(d1[k][0], d3[k][0]), k
    # the survivors: an independent kernel family counts them again
    sel, fst = run(columns, COND, [], [], [(0, ROWS)], filter_only=True)
    assert st["rows_selected"] == fst["rows_selected"] == len(sel["selected"])
    # unequal batches through the same specialised kernel (three calls: reset / push x 4 / finish)
    cuts = [0, 24_999_999, 50_000_001, 75_000_003, ROWS]
    parts, pst = run(columns, COND, KEYS, aggs, list(zip(cuts[:-1], cuts[1:])))
    assert pst["spec_kernel"] == 1 and pst["rows_selected"] == st["rows_selected"]
    dp = as_dict(parts)
    for k in d1:
        assert close(d1[k][0], dp[k][0]), (k, d1[k], dp[k])
    # another kernel family over the same rows: the 4-aggregate plan of test_full_size_properties (bounded-shape kernel)
    other, ost = run(columns, COND, KEYS, AGGS, [(0, ROWS)])
    assert ost["spec_kernel"] == 0
    do, si = as_dict(other), AGGS.index("sum(%s)" % D("price"))
    for k in d1:
        assert close(d1[k][0], do[k][si]), (k, d1[k], do[k][si])
    # and the oracle on a prefix of the same data set, through the same kernel
    n = 2_000_000
    t = n1o.synth_table(n, k_cat=K_CAT, total_rows=ROWS)
    ora = n1o.run(t, COND, KEYS, aggs, threads=4)
    raw, rst = run(columns, COND, KEYS, aggs, [(0, n)])
    assert rst["spec_kernel"] == 1 and rst["rows_selected"] == ora.rows_passed
    op = query_amd.GpuFilterGroup(plan.filter_group_plan(COND, KEYS, aggs))
    op.intern(bench.synth_dictionary(K_CAT))
    cache = {}
    from query_amd.gpu_operator import GroupRows
    got = GroupRows(1, 1, op._py_values(raw["keys"], cache), op._py_values(raw["aggs"], cache), [])
    op.done()
    pu.assert_same_groups(got, ora, aggs=aggs)


def test_full_size_prefix_matches_oracle(columns):
    """The first 2 M rows of the 100 M-row device data set through the oracle (same generator on the CPU)."""
    n = 2_000_000
    t = n1o.synth_table(n, k_cat=K_CAT, total_rows=ROWS)
    ora = n1o.run(t, COND, KEYS, AGGS, threads=4)
    raw, st = run(columns, COND, KEYS, AGGS, [(0, n)])
    assert st["rows_selected"] == ora.rows_passed
    op = query_amd.GpuFilterGroup(plan.filter_group_plan(COND, KEYS, AGGS))
    import bench
    op.intern(bench.synth_dictionary(K_CAT))
    cache = {}
    from query_amd.gpu_operator import GroupRows
    got = GroupRows(1, len(AGGS), op._py_values(raw["keys"], cache), op._py_values(raw["aggs"], cache), [])
    op.done()
    pu.assert_same_groups(got, ora, aggs=AGGS)


def test_full_size_count_distinct_properties(columns):
    """COUNT(DISTINCT user_id) at 100 M rows: batching invariance and the bound distinct <= count <= rows."""
    aggs = sorted(["count(distinct %s)" % D("user_id"), "count(%s)" % D("user_id")])
    whole, _ = run(columns, None, KEYS, aggs, [(0, ROWS)])
    halves, _ = run(columns, None, KEYS, aggs, [(0, ROWS // 2 + 1), (ROWS // 2 + 1, ROWS)])
    a, b = as_dict(whole), as_dict(halves)
    assert a == b
    di, ci = aggs.index("count(distinct %s)" % D("user_id")), aggs.index("count(%s)" % D("user_id"))
    assert sum(v[ci][1] for v in a.values()) == ROWS
    assert all(0 < v[di][1] <= v[ci][1] for v in a.values())


def test_full_size_count_distinct_paths_agree(columns):
    """The two independent ways the sets are built — radix partition + LDS sets over one-word members, and the
    per-group global sets over (key, value, class) pairs — give the same COUNT(DISTINCT) at 100 M rows; the oracle
    pins both on a prefix."""
    aggs = sorted(["count(distinct %s)" % D("user_id"), "avg(%s)" % D("price")])  # BASELINE config 3
    words, ws = run(columns, None, KEYS, aggs, [(0, ROWS)])
    pairs, ps = run(columns, None, KEYS, aggs, [(0, ROWS)], distinct_words=0)
    assert ws["distinct_path"] == 2 and ps["distinct_path"] == 1
    a, b = as_dict(words), as_dict(pairs)
    assert a.keys() == b.keys() and len(a) == K_CAT
    for k in a:
        for x, y in zip(a[k], b[k]):
            assert close(x, y), (k, x, y)
    n = 1_000_000
    t = n1o.synth_table(n, k_cat=K_CAT, total_rows=ROWS)
    ora = n1o.run(t, None, KEYS, aggs, threads=4)
    raw, _ = run(columns, None, KEYS, aggs, [(0, n)])
    import bench
    op = query_amd.GpuFilterGroup(plan.filter_group_plan(None, KEYS, aggs))
    op.intern(bench.synth_dictionary(K_CAT))
    cache = {}
    from query_amd.gpu_operator import GroupRows
    got = GroupRows(1, len(aggs), op._py_values(raw["keys"], cache), op._py_values(raw["aggs"], cache), [])
    op.done()
    pu.assert_same_groups(got, ora, aggs=aggs)


def test_full_size_high_cardinality_paths_agree(columns):
    """GROUP BY user_id (10 M groups in 100 M rows): the partitioned path (records -> radix passes -> per-bin LDS
    tables) against the scan kernel with global atomics, group by group; COUNT(*) adds up to the rows."""
    keys = [D("user_id")]
    aggs = sorted(["count(*)", "sum(%s)" % D("region_id"), "max(%s)" % D("price")])
    part, pst = run(columns, None, keys, aggs, [(0, ROWS)])
    assert pst["agg_mode"] == 4  # chosen from the data: the first rows bring far more than 65 536 groups
    scan, sst = run(columns, None, keys, aggs, [(0, ROWS)], agg_mode=_ffi.MODE_LDS_HASH)
    assert sst["agg_mode"] != 4
    # 100 M uniform draws over 10 M ids leave about 10 M * e^-10 = 454 ids unseen
    assert part["ngroups"] == scan["ngroups"] and ROWS // 10 - 1000 < part["ngroups"] <= ROWS // 10

    def table(raw):
        order = np.argsort(raw["keys"][:, 0]["v"], kind="stable")
        return raw["keys"][order], raw["aggs"][order]

    pk, pa = table(part)
    sk, sa = table(scan)
    assert np.array_equal(pk["v"], sk["v"]) and np.array_equal(pk["tag"], sk["tag"])
    assert np.array_equal(pa["tag"], sa["tag"])
    assert np.array_equal(pa["v"], sa["v"])  # COUNT, integer SUM and MAX are bit-exact whatever the order of the rows
    ci = aggs.index("count(*)")
    assert int(pa[:, ci]["v"].astype(np.int64).sum()) == ROWS


def test_full_size_config5_own_query():
    """BASELINE config 5's own query — GROUP BY cat, region_id (K_cat = 100 000: 6.4 M groups) ORDER BY SUM(price) DESC
    LIMIT 100 — at 100 M rows: the partitioned path with the plan-specialised records front end (chosen from the data)
    against the scan-kernel path (LDS hash + global table) and the three-array partitioned path, rank by rank and, without
    the tail, group by group; then the same query on a prefix of the same data set against the oracle."""
    import bench
    k5 = 100_000
    cols = bench.DeviceColumns(ROWS, k5, False, 0, ROWS, 0)
    keys = [D("cat"), D("region_id")]
    aggs = ["sum(%s)" % D("price")]
    order = [(aggs[0], True)]

    def f64(v):
        return v.astype(np.uint64).view(np.float64)

    def near(a, b):
        return np.abs(a - b) <= 1e-9 * np.maximum(np.abs(a), np.abs(b))

    # --- the tail: 100 rows leave the device
    rec, rst = run(cols, None, keys, aggs, [(0, ROWS)], kcat=k5, order=order, limit=100)
    assert rst["agg_mode"] == 4 and rst["spec_kernel"] != 0  # partitioned, records written by the specialised scan
    assert rec["ngroups"] == 100 and rst["rows_selected"] == ROWS
    scan, sst = run(cols, None, keys, aggs, [(0, ROWS)], kcat=k5, order=order, limit=100, agg_mode=_ffi.MODE_LDS_HASH)
    assert sst["agg_mode"] != 4
    old, ost = run(cols, None, keys, aggs, [(0, ROWS)], kcat=k5, order=order, limit=100, records=0)
    assert ost["agg_mode"] == 4 and ost["spec_kernel"] == 0
    for other in (scan, old):
        assert other["ngroups"] == 100
        a, b = rec["aggs"][:, 0], other["aggs"][:, 0]
        assert np.array_equal(a["tag"], b["tag"]) and np.all(a["tag"] == _ffi.T_FLOAT)
        sa, sb = f64(a["v"]), f64(b["v"])
        assert np.all(near(sa, sb)), "rank sums differ"
        assert np.all(sa[:-1] >= sa[1:] * (1 - 1e-9))  # descending
        # rows may trade places only where neighbouring sums agree within the float tolerance
        same = np.all(rec["keys"]["v"] == other["keys"]["v"], axis=1)
        for i in np.nonzero(~same)[0]:
            lo, hi = max(i - 1, 0), min(i + 1, 99)
            assert near(sa[i], sa[lo]) or near(sa[i], sa[hi]), ("row %d differs beyond ties" % i)
    # --- without the tail: all groups, records front end vs three-array records
    rec, rst = run(cols, None, keys, aggs, [(0, ROWS)], kcat=k5)
    old, ost = run(cols, None, keys, aggs, [(0, ROWS)], kcat=k5, records=0)
    assert rst["agg_mode"] == 4 and rst["spec_kernel"] != 0 and ost["spec_kernel"] == 0
    # 100 M uniform draws over 6.4 M keys leave 6.4 M * e^-15.6 ~ 1 key unseen
    assert rec["ngroups"] == old["ngroups"] and 64 * k5 - 100 < rec["ngroups"] <= 64 * k5

    def table(raw):
        k = raw["keys"]["v"].astype(np.uint64)
        order = np.argsort(k[:, 0] * np.uint64(64) + k[:, 1], kind="stable")
        return raw["keys"][order], raw["aggs"][order]

    rk, ra = table(rec)
    ok, oa = table(old)
    assert np.array_equal(rk["v"], ok["v"]) and np.array_equal(rk["tag"], ok["tag"])
    assert np.array_equal(ra["tag"], oa["tag"])
    isf = ra["tag"][:, 0] == _ffi.T_FLOAT
    assert np.array_equal(ra["v"][~isf], oa["v"][~isf])  # integer sums are bit-exact
    assert np.all(near(f64(ra["v"][isf]), f64(oa["v"][isf])))
    del cols
    # --- the oracle on a prefix of the same data set, the partitioned path forced (a prefix is below the probe's threshold)
    n = 400_000
    t = n1o.synth_table(n, k_cat=k5, total_rows=ROWS)
    ora = n1o.run(t, None, keys, aggs, threads=4)
    gpu, st = pu.run_gpu(t, None, keys, aggs, device_resident=True, agg_mode=4)
    assert st["agg_mode"] == 4 and st["spec_kernel"] != 0
    pu.assert_same_groups(gpu, ora, aggs=aggs)
    gpu, st = pu.run_gpu(t, None, keys, aggs, device_resident=True, agg_mode=4, order=order, limit=100, topk_min_groups=1024)
    assert st["agg_mode"] == 4
    pu.assert_ordered_groups(gpu, ora, keys, aggs, order, limit=100)
