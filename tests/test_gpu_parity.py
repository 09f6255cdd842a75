"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the reference's golden cases."""
import os

import numpy as np
import pytest

import golden_util as gu
import parity_util as pu
import query_amd
from oracle import n1o
from query_amd import _ffi

pytestmark = pytest.mark.gpu

ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = gu.load_cases()


def D(*names):
    return query_amd.plan.field_path("default", *names)


@pytest.mark.parametrize("case", CASES, ids=[c["id"] for c in CASES])
def test_golden_cases_on_device(case):
    """The reference's own expected results, produced by the HIP path (or a clean N1K_UNSUPPORTED)."""
    docs = gu.load_docs(case["keyspace"])
    plan = case["plan"]
    try:
        if "exprs" in plan:
            # constant expressions (case_integer.json) as group keys of a one-row table: the device's arithmetic
            one = gu.build_table(docs[:1], [])
            got = [{}]
            for i in range(0, len(plan["exprs"]), 4):
                part = plan["exprs"][i:i + 4]
                rows, _ = pu.run_gpu(one, None, [text for _a, text in part], ["count(*)"])
                assert len(rows.keys) == 1
                got[0].update({alias: gu.decode_value(tv) for (alias, _t), tv in zip(part, rows.keys[0])})
            assert gu.same_json(got, case["results"]), (got, case["results"])
            return
        if "row_expr" in plan:
            # one value per document (case_func_num.json): the expression as the operand of MAX, one group per document —
            # the device's arith_apply (derived column here; test_fused_* run the same code inside the scan)
            alias, text = plan["row_expr"]
            idp = "(`game`.`id`)"
            table = gu.build_table(docs, gu.leaf_paths({"condition": None, "group_keys": [idp, text], "aggregates": []}))
            rows, _ = pu.run_gpu(table, None, [idp], ["max(%s)" % text])
            assert len(rows.keys) == len(docs)
            got = gu.sorted_values(alias, [gu.decode_value(a[0]) for a in rows.aggs])
            assert gu.same_json(got, case["results"]), (got, case["results"])
            return
        if "row_expr_by" in plan:
            # one value per document in the order of another field (case_func_comp.json): one group per value of that field
            alias, text, by = plan["row_expr_by"]
            table = gu.build_table(docs, gu.leaf_paths({"condition": None, "group_keys": [by, text], "aggregates": []}))
            rows, _ = pu.run_gpu(table, None, [by], ["max(%s)" % text])
            assert len(rows.keys) == len(docs)
            got = gu.values_ordered_by(alias, [gu.decode_value(a[0]) for a in rows.aggs], [gu.decode_value(k[0]) for k in rows.keys])
            assert gu.same_json(got, case["results"]), (got, case["results"])
            return
        table = gu.build_table(docs, gu.leaf_paths(plan))
        if plan.get("filter_only"):
            rows, _ = pu.run_gpu(table, plan["condition"], [], [], filter_only=True)
            got = gu.replay_filter_post(case, docs, rows.selected)
        else:
            # the whole grouped tail runs inside the handle: HAVING through the device's predicate evaluator, the SELECT
            # list as InitialProject (ROUND on the device), ORDER BY / LIMIT over keys, aggregates and projection aliases
            rows, _ = pu.run_gpu(table, plan["condition"], plan["group_keys"], plan["aggregates"], having=gu.having_text(case),
                                 project=gu.project_terms(case), order=gu.order_terms(case), limit=case["post"].get("limit"))
            got = gu.rows_from_projection(case, rows)
            # ... and the groups alone, replayed by the harness, agree with it
            plain, _ = pu.run_gpu(table, plan["condition"], plan["group_keys"], plan["aggregates"], having=gu.having_text(case))
            assert gu.same_json(gu.replay_post(case, gu.groups_from_result(plain), having_done=True), case["results"])
    except query_amd.N1kError as e:
        if e.status == _ffi.UNSUPPORTED:
            pytest.skip("outside the device subset: " + e.message)
        raise
    assert gu.same_json(got, case["results"]), (got, case["results"])


CONFIG2 = ("(50 < %s)" % D("price"), [D("cat")], ["sum(%s)" % D("price")])


@pytest.mark.parametrize("k_cat,zipf", [(1000, False), (16, False), (1000, True), (3, True)])
@pytest.mark.parametrize("device_resident", [False, True])
def test_config2_filter_group_sum(k_cat, zipf, device_resident):
    t = n1o.synth_table(200_000, k_cat=k_cat, zipf=zipf)
    cond, keys, aggs = CONFIG2
    ora = n1o.run(t, cond, keys, aggs, threads=2)
    gpu, stats = pu.run_gpu(t, cond, keys, aggs, device_resident=device_resident)
    pu.assert_same_groups(gpu, ora, aggs=aggs)
    assert stats["rows_in"] == t.nrows
    assert stats["rows_selected"] == ora.rows_passed  # bit-exact COUNT of the filter


SPEC_SHAPES = {
    "gt_sum": ("(50 < %s)" % D("price"), [D("cat")], ["sum(%s)" % D("price")]),
    "lt_sum": ("(%s < 30)" % D("price"), [D("cat")], ["sum(%s)" % D("price")]),
    "gtf_sum": ("(%s > 49.5)" % D("price") if False else "(49.5 < %s)" % D("price"), [D("cat")], ["sum(%s)" % D("price")]),
    "gt_all": ("(50 < %s)" % D("price"), [D("cat")],
               sorted(["avg(%s)" % D("price"), "count(*)", "max(%s)" % D("price"), "min(%s)" % D("price"),
                       "sum(%s)" % D("price")])),
    "gt_count": ("(50 < %s)" % D("price"), [D("cat")], ["count(*)"]),
    "sum": (None, [D("cat")], ["sum(%s)" % D("price")]),
    "avg": (None, [D("cat")], ["avg(%s)" % D("price")]),
    "count": (None, [D("cat")], ["count(*)"]),
    "gt_nokey_count": ("(50 < %s)" % D("price"), [], ["count(*)"]),
    # integer keys: open-addressed LDS table inside the specialised kernel
    "ik_count": (None, [D("region_id")], ["count(*)"]),
    "ik_sum": (None, [D("region_id")], ["sum(%s)" % D("price")]),
    "dik_sum": (None, [D("cat"), D("region_id")], ["sum(%s)" % D("price")]),
    "2k_sum": (None, [D("cat"), D("cat")], ["sum(%s)" % D("price")]) if False else (None, [D("cat")], ["sum(%s)" % D("price")]),
}


@pytest.mark.parametrize("shape", sorted(SPEC_SHAPES))
@pytest.mark.parametrize("opts", [{}, {"wide": 0}, {"spec": 0}, {"fast": 0}, {"fast": 0, "agg_mode": 1}, {"block": 512}],
                         ids=["spec", "spec-narrow", "fast", "interp-direct", "interp-hash", "spec-b512"])
def test_every_kernel_variant_agrees_with_the_oracle(shape, opts):
    """The plan-specialised, the bounded-shape and the interpreted kernels are three implementations of one
    semantics: all must match the oracle (odd row count: exercises the tail of the 2-rows-per-lane loads)."""
    cond, keys, aggs = SPEC_SHAPES[shape]
    t = n1o.synth_table(150_001, k_cat=37, zipf=True)
    ora = n1o.run(t, cond, keys, aggs)
    gpu, stats = pu.run_gpu(t, cond, keys, aggs, device_resident=True, **opts)
    pu.assert_same_groups(gpu, ora, aggs=aggs)
    assert stats["rows_selected"] == ora.rows_passed


def test_hashed_spec_kernel_with_more_groups_than_lds_slots():
    """Integer key with far more groups than LDS slots: rows whose group does not fit the workgroup table take the
    global path; results stay exact."""
    n = 300_000
    t = n1o.synth_table(n, k_cat=7, total_rows=3_000_000)  # user_id in [0, 300000): ~190k groups
    keys, aggs = [D("user_id")], ["sum(%s)" % D("price")]
    ora = n1o.run(t, None, keys, aggs)
    gpu, stats = pu.run_gpu(t, None, keys, aggs, device_resident=True)
    assert stats["spec_kernel"] == 1
    pu.assert_same_groups(gpu, ora, aggs=aggs)


JIT_SHAPES = [
    ("((10 < %s) and (%s <= 90.5))" % (D("price"), D("price")), [D("cat")],
     ["avg(%s)" % D("price"), "count(*)", "max(%s)" % D("user_id")]),
    ("(%s is not null)" % D("price"), [D("cat")], ["min(%s)" % D("price"), "sum(%s)" % D("region_id")]),
    ("(%s = \"cat_3\")" % D("cat"), [D("cat")], ["count(%s)" % D("price"), "countn(%s)" % D("price")]),
    ("(%s < 30)" % D("price"), [D("region_id")], ["count(*)", "min(%s)" % D("price")]),
    (None, [], ["sum(%s)" % D("price"), "avg(%s)" % D("user_id")]),
    ("(50 <= %s)" % D("price"), [D("cat"), D("region_id")], ["max(%s)" % D("price")]),
]


@pytest.mark.parametrize("case", range(len(JIT_SHAPES)))
def test_runtime_specialised_kernels_agree_with_the_oracle(case):
    """Shapes without an ahead-of-time kernel: the scan template is instantiated at run time (hiprtc) when forced
    (jit=2; by default only large batches trigger it).  Same semantics, checked against the oracle."""
    cond, keys, aggs = JIT_SHAPES[case]
    aggs = sorted(aggs)
    t = n1o.synth_table(90_001, k_cat=23, zipf=True)
    ora = n1o.run(t, cond, keys, aggs)
    gpu, stats = pu.run_gpu(t, cond, keys, aggs, device_resident=True, jit=2)
    assert stats["spec_kernel"] == 2, "the run-time instantiation did not run"
    pu.assert_same_groups(gpu, ora, aggs=aggs)
    assert stats["rows_selected"] == ora.rows_passed
    gpu2, stats2 = pu.run_gpu(t, cond, keys, aggs, device_resident=False, batches=3, jit=2)
    pu.assert_same_groups(gpu2, ora, aggs=aggs)


def test_unaligned_device_columns():
    """Columns that start at an odd row of a larger allocation cannot use 16-byte loads: same results."""
    import torch
    cond, keys, aggs = SPEC_SHAPES["gt_sum"]
    t = n1o.synth_table(50_001, k_cat=20)
    sub = t.slice(1, 50_001)
    ora = n1o.run(sub, cond, keys, aggs)
    op = query_amd.GpuFilterGroup(query_amd.plan.filter_group_plan(cond, keys, aggs))
    op.intern(list(t.dictionary))
    by = {c.name: c for c in t.columns}
    keep, dev = [], []
    for p in op.column_paths:
        c = by[p]
        if c.kind == n1o.COL_DICT32:
            x = torch.from_numpy(c.codes.view(np.int32)).cuda()
            keep.append(x)
            dev.append((_ffi.COL_DICT32, None, None, x.data_ptr() + 4))
        else:
            a = torch.from_numpy(c.tags).cuda()
            b = torch.from_numpy(c.payload.view(np.int64)).cuda()
            keep += [a, b]
            dev.append((_ffi.COL_TAGGED64, a.data_ptr() + 1, b.data_ptr() + 8, None))
    torch.cuda.synchronize()
    op.process_device_items(50_000, dev)
    gpu = op.after_items()
    op.done()
    pu.assert_same_groups(gpu, ora, aggs=aggs)


ALL_AGGS = sorted(["sum(%s)" % D("price"), "avg(%s)" % D("price"), "min(%s)" % D("price"), "max(%s)" % D("price"),
                   "count(*)", "count(%s)" % D("price"), "countn(%s)" % D("price"), "sum(%s)" % D("user_id")])


@pytest.mark.parametrize("cond", [None, "(50 < %s)" % D("price"), "(%s <= 20.5)" % D("price"),
                                  "((10 < %s) and (%s < 90))" % (D("price"), D("price")),
                                  "((%s < 5) or (%s is null) or (%s is missing))" % (D("price"), D("price"), D("price")),
                                  "(not (%s between 25 and 75))" % D("price"),
                                  "(%s = \"n/a\")" % D("price"), "(%s is valued)" % D("price"),
                                  "(%s = \"cat_7\")" % D("cat")])
def test_all_aggregates_many_filters(cond):
    t = n1o.synth_table(120_000, k_cat=40)
    ora = n1o.run(t, cond, [D("cat")], ALL_AGGS)
    gpu, _ = pu.run_gpu(t, cond, [D("cat")], ALL_AGGS)
    pu.assert_same_groups(gpu, ora, aggs=ALL_AGGS)


@pytest.mark.parametrize("keys", [[], [D("region_id")], [D("cat"), D("region_id")], [D("price")], [D("user_id")]])
def test_key_shapes(keys):
    n = 60_000
    t = n1o.synth_table(n, k_cat=50)
    if keys == [D("price")]:
        # non-integral float keys: grouped through the wide-value tables (about 10 k distinct prices)
        ora = n1o.run(t, None, keys, ["count(*)"])
        gpu, stats = pu.run_gpu(t, None, keys, ["count(*)"])
        pu.assert_same_groups(gpu, ora)
        assert stats["wide_key_values"] > 5000
        return
    aggs = sorted(["count(*)", "sum(%s)" % D("price"), "max(%s)" % D("user_id")])
    ora = n1o.run(t, "(%s is not missing)" % D("price"), keys, aggs)
    gpu, _ = pu.run_gpu(t, "(%s is not missing)" % D("price"), keys, aggs, batches=3)
    pu.assert_same_groups(gpu, ora, aggs=aggs)


def test_empty_input_default_row():
    t = n1o.synth_table(1000, k_cat=5)
    cond = "(%s < -1)" % D("price")  # nothing passes (strings sort above numbers, so use <)
    aggs = sorted(["count(*)", "sum(%s)" % D("price"), "min(%s)" % D("price"), "avg(%s)" % D("price")])
    ora = n1o.run(t, cond, [], aggs)
    gpu, _ = pu.run_gpu(t, cond, [], aggs)
    assert len(ora.keys) == 1  # FinalGroup's default row (execution/group_final.go:108-117)
    pu.assert_same_groups(gpu, ora, aggs=aggs)
    ora = n1o.run(t, cond, [D("cat")], aggs)
    gpu, _ = pu.run_gpu(t, cond, [D("cat")], aggs)
    assert len(ora.keys) == 0 and len(gpu.keys) == 0


def test_int64_exact_sum_and_sign_rule():
    """intValue.Add keeps int64 only for same-sign operands without overflow (value/integer.go:266-277)."""
    n = 4096
    rng = np.random.default_rng(7)
    grp = rng.integers(0, 8, n).astype(np.uint64)
    vals = np.zeros(n, dtype=np.int64)
    big = 9223372036854775807 // 1024
    for g in range(8):
        idx = np.nonzero(grp == g)[0]
        if g == 0:
            vals[idx] = rng.integers(big - 1000, big, len(idx))        # large positives, no overflow
        elif g == 1:
            vals[idx] = -rng.integers(1, 1 << 45, len(idx))            # all negative: stays int
        elif g == 2:
            vals[idx] = rng.integers(-100, 100, len(idx))              # mixed signs: float per the reference
        elif g == 3:
            vals[idx] = 9223372036854775807 // 4                       # overflows int64: float
        else:
            vals[idx] = rng.integers(0, 1 << 50, len(idx))
    tags = np.full(n, n1o.T_INT, np.uint8)
    t = n1o.Table([n1o.Column(D("g"), n1o.COL_TAGGED64, tags=tags.copy(), payload=grp),
                   n1o.Column(D("v"), n1o.COL_TAGGED64, tags=tags.copy(), payload=vals.view(np.uint64))], [])
    aggs = sorted(["sum(%s)" % D("v"), "avg(%s)" % D("v"), "min(%s)" % D("v"), "max(%s)" % D("v")])
    ora = n1o.run(t, None, [D("g")], aggs)
    gpu, _ = pu.run_gpu(t, None, [D("g")], aggs)
    pu.assert_same_groups(gpu, ora, aggs=aggs)
    kinds = {k[0][1]: a[aggs.index("sum(%s)" % D("v"))][0] for k, a in zip(gpu.keys, gpu.aggs)}
    assert kinds[0] == n1o.T_INT and kinds[1] == n1o.T_INT and kinds[2] == n1o.T_FLOAT and kinds[3] == n1o.T_FLOAT


CONFIG3 = (None, [D("cat")], sorted(["count(distinct %s)" % D("user_id"), "avg(%s)" % D("price")]))


@pytest.mark.parametrize("batches", [1, 3])
def test_config3_count_distinct_avg(batches):
    """BASELINE config 3: COUNT(DISTINCT user_id) + AVG(price) GROUP BY cat (exact set semantics, value/set.go)."""
    t = n1o.synth_table(200_000, k_cat=100, total_rows=20_000)  # user_id in [0, 2000): many duplicates per group
    cond, keys, aggs = CONFIG3
    ora = n1o.run(t, cond, keys, aggs, threads=2)
    gpu, _ = pu.run_gpu(t, cond, keys, aggs, batches=batches, device_resident=(batches == 1))
    pu.assert_same_groups(gpu, ora, aggs=aggs)


def test_distinct_over_mixed_types_and_edge_values():
    """Integral floats join the ints, -1 collides with the free marker, strings/booleans count by value,
    NULL/MISSING never enter the set; COUNTN(DISTINCT) takes numbers only."""
    n = 6000
    rng = np.random.default_rng(11)
    grp = rng.integers(0, 5, n).astype(np.uint64)
    tags = np.zeros(n, np.uint8)
    pay = np.zeros(n, np.uint64)
    strs = [b"a", b"b", b"", b"zz"]
    for i in range(n):
        r = rng.integers(0, 10)
        if r < 3:
            tags[i], pay[i] = n1o.T_INT, np.int64(rng.integers(-3, 4)).view(np.uint64)
        elif r < 5:
            f = float(rng.integers(-3, 4)) if rng.integers(0, 2) else float(rng.integers(0, 8)) / 4.0 + 0.125
            tags[i], pay[i] = n1o.T_FLOAT, np.float64(f).view(np.uint64)  # some integral floats (unfolded on purpose)
        elif r < 7:
            tags[i], pay[i] = n1o.T_STRING, rng.integers(0, len(strs))
        elif r == 7:
            tags[i] = n1o.T_TRUE if rng.integers(0, 2) else n1o.T_FALSE
        elif r == 8:
            tags[i] = n1o.T_NULL
        else:
            tags[i] = n1o.T_MISSING
    gt = np.full(n, n1o.T_INT, np.uint8)
    t = n1o.Table([n1o.Column(D("g"), n1o.COL_TAGGED64, tags=gt, payload=grp),
                   n1o.Column(D("v"), n1o.COL_TAGGED64, tags=tags, payload=pay)], strs)
    aggs = sorted(["count(distinct %s)" % D("v"), "countn(distinct %s)" % D("v"), "count(%s)" % D("v"),
                   "sum(distinct %s)" % D("v"), "avg(distinct %s)" % D("v")])
    for keys in ([D("g")], []):
        ora = n1o.run(t, None, keys, aggs)
        gpu, _ = pu.run_gpu(t, None, keys, aggs, batches=2)
        pu.assert_same_groups(gpu, ora, aggs=aggs)


ARITH_CASES = [
    # (condition, keys, aggregates)
    ("(100 < (%s + %s))" % (D("price"), D("region_id")), [D("cat")], ["count(*)", "sum((%s * %s))" % (D("price"), D("region_id"))]),
    ("((%s * 2) < 51)" % D("price"), [D("cat")], ["avg((%s - 10))" % D("price"), "max((-%s))" % D("price")]),
    (None, ["(%s %% 4)" % D("region_id")], ["sum((%s / 4))" % D("price"), "min((%s / (%s - 7)))" % (D("price"), D("region_id"))]),
    (None, ["idiv(%s, 1000)" % D("user_id"), "imod(%s, 3)" % D("region_id")], ["count(*)", "sum((%s + %s + %s + %s + %s))" % ((D("region_id"),) * 5)]),
    ("(((%s + 1) * (%s + 1)) between 100 and 2000)" % (D("region_id"), D("price")), [], ["count(*)", "countn((%s + %s))" % (D("price"), D("cat"))]),
]


@pytest.mark.parametrize("case", range(len(ARITH_CASES)))
def test_arithmetic_operands_as_derived_columns(case):
    """expression/arith_*.go on the device: arithmetic nodes become derived columns (element-wise kernel per node)."""
    cond, keys, aggs = ARITH_CASES[case]
    aggs = sorted(aggs)
    t = n1o.synth_table(70_000, k_cat=12)
    ora = n1o.run(t, cond, keys, aggs)
    gpu, _ = pu.run_gpu(t, cond, keys, aggs, batches=2)
    pu.assert_same_groups(gpu, ora, aggs=aggs)


# which of ARITH_CASES the run-time-built scan takes with its arithmetic in registers (<= 3 nodes over <= 3 input columns,
# a bounded shape): the others keep their derived columns
ARITH_FUSED = {0: True, 1: True, 2: False, 3: False, 4: False}


@pytest.mark.parametrize("case", range(len(ARITH_CASES)))
@pytest.mark.parametrize("resident", [False, True])
def test_arithmetic_fused_into_the_runtime_built_scan(case, resident):
    """expression/arith_*.go evaluated in registers by the run-time-built plan-specialised scan (no derived column in
    HBM): same groups as the oracle and as the derived-column path, in one batch and in two (odd sizes: the narrow tail)."""
    cond, keys, aggs = ARITH_CASES[case]
    aggs = sorted(aggs)
    t = n1o.synth_table(70_001, k_cat=12)
    ora = n1o.run(t, cond, keys, aggs)
    for batches in (1, 2):
        gpu, st = pu.run_gpu(t, cond, keys, aggs, batches=batches, device_resident=resident, jit=2)
        pu.assert_same_groups(gpu, ora, aggs=aggs)
        assert (st["spec_kernel"] == 3) == ARITH_FUSED[case], st
    unfused, st = pu.run_gpu(t, cond, keys, aggs, batches=2, device_resident=resident, jit=2, fuse_arith=0)
    pu.assert_same_groups(unfused, ora, aggs=aggs)
    assert st["spec_kernel"] != 3


@pytest.mark.parametrize("opts", [{}, {"jit": 2}], ids=["derived-columns", "fused"])
def test_int64_multiplication_overflow_corners(opts):
    """intValue.Mult (value/integer.go:318-329) keeps the int64 when `x == 0 || rv / x == y`, else multiplies as floats;
    the device decides from the high half of the 128-bit product.  Every pair of corner values, one row per group."""
    corners = [0, 1, -1, 2, -2, 3, 7, -7, 2 ** 31, -(2 ** 31), 2 ** 32, 2 ** 32 + 1, 3037000499, 3037000500, -3037000500,
               2 ** 62, -(2 ** 62), 2 ** 63 - 1, -(2 ** 63), -(2 ** 63) + 1, 4611686018427387904, 6148914691236517205]
    xs = np.array([a for a in corners for _ in corners], dtype=object)
    ys = np.array([b for _ in corners for b in corners], dtype=object)
    n = len(xs)
    as_u64 = lambda v: np.array([int(t) & (2 ** 64 - 1) for t in v], dtype=np.uint64)
    ints = np.full(n, n1o.T_INT, np.uint8)
    t = n1o.Table([n1o.Column(D("id"), n1o.COL_TAGGED64, tags=ints, payload=np.arange(n, dtype=np.uint64)),
                   n1o.Column(D("x"), n1o.COL_TAGGED64, tags=ints, payload=as_u64(xs)),
                   n1o.Column(D("y"), n1o.COL_TAGGED64, tags=ints, payload=as_u64(ys))], [])
    aggs = ["max((%s * %s))" % (D("x"), D("y"))]
    ora = n1o.run(t, None, [D("id")], aggs)
    gpu, _ = pu.run_gpu(t, None, [D("id")], aggs, **opts)
    pu.assert_same_groups(gpu, ora, aggs=aggs)
    got = dict(zip([k[0][1] for k in gpu.keys], [a[0] for a in gpu.aggs]))
    for i in range(n):
        exact = int(xs[i]) * int(ys[i])
        fits = -(2 ** 63) <= exact < 2 ** 63 or (int(xs[i]) == -1 and int(ys[i]) == -(2 ** 63))
        assert (got[i][0] == n1o.T_INT) == fits, (int(xs[i]), int(ys[i]), got[i])


FUSED_EXTRA = [
    # ROUND and IDIV nodes, a node over a node, a computed (hashed) key, string operands (NULL)
    ("(100 <= (%s * 3))" % D("price"), [D("cat")], ["count(*)", "sum(round((%s * 1.5), 1))" % D("price")]),
    (None, [D("cat")], ["avg((%s / %s))" % (D("price"), D("region_id")), "countn((%s + %s))" % (D("price"), D("cat"))]),
    ("(idiv(%s, 7) = 3)" % D("price"), [D("cat")], ["max((%s %% 5))" % D("user_id"), "min((-%s))" % D("user_id")]),
    (None, ["(%s + 1)" % D("region_id")], ["sum((%s * %s))" % (D("price"), D("price")), "count(*)"]),
]


@pytest.mark.parametrize("case", range(len(FUSED_EXTRA)))
def test_fused_arithmetic_directed_shapes(case):
    cond, keys, aggs = FUSED_EXTRA[case]
    aggs = sorted(aggs)
    t = n1o.synth_table(50_003, k_cat=9)
    ora = n1o.run(t, cond, keys, aggs)
    gpu, st = pu.run_gpu(t, cond, keys, aggs, batches=2, jit=2)
    pu.assert_same_groups(gpu, ora, aggs=aggs)
    assert st["spec_kernel"] == 3, st


@pytest.mark.parametrize("tail_in_merge", [0, 1], ids=["merge-then-tail", "tail-in-the-merge"])
def test_run_device_batch_is_reset_push_finish(tail_in_merge):
    """n1k_run_device_batch = n1k_reset + n1k_push_device_batch + n1k_finish: same groups as the three calls, every time —
    with the query's tail as its own kernel and (option tail_in_merge) run by the merge kernel's last workgroup."""
    import torch
    cond, keys = "(50 < %s)" % D("price"), [D("cat")]
    aggs = sorted(["sum(%s)" % D("price"), "count(*)", "min(%s)" % D("user_id")])
    t = n1o.synth_table(200_003, k_cat=40)
    ora = n1o.run(t, cond, keys, aggs)
    op = query_amd.GpuFilterGroup(query_amd.plan.filter_group_plan(cond, keys, aggs), tail_in_merge=tail_in_merge)
    op.intern(list(t.dictionary))
    by = {c.name: c for c in t.columns}
    keep, cols = [], []
    for p in op.column_paths:
        c = by[p]
        if c.kind == n1o.COL_DICT32:
            x = torch.from_numpy(c.codes.view(np.int32)).cuda()
            keep.append(x)
            cols.append((_ffi.COL_DICT32, None, None, x.data_ptr()))
        else:
            a, b = torch.from_numpy(c.tags).cuda(), torch.from_numpy(c.payload.view(np.int64)).cuda()
            keep += [a, b]
            cols.append((_ffi.COL_TAGGED64, a.data_ptr(), b.data_ptr(), None))
    torch.cuda.synchronize()
    batch = op.make_device_batch(t.nrows, cols)
    from query_amd.gpu_operator import GroupRows
    for _ in range(3):
        raw = op.run_device_batch_raw(batch)
        cache = {}
        got = GroupRows(len(keys), len(aggs), op._py_values(raw["keys"], cache), op._py_values(raw["aggs"], cache), [])
        pu.assert_same_groups(got, ora, aggs=aggs)
        assert op.stats()["rows_selected"] == ora.rows_passed
    op.reopen()
    op.process_device_batch(batch)
    three = op.after_items()
    pu.assert_same_groups(three, ora, aggs=aggs)
    op.done()


@pytest.mark.parametrize("stream", [1, 0], ids=["one-pass", "mask-scan-compact"])
def test_filter_only_selected_rows(stream):
    t = n1o.synth_table(100_003, k_cat=10)
    for cond in ["(50 < %s)" % D("price"), "(%s is missing)" % D("price"), "(%s = \"cat_3\")" % D("cat")]:
        ora = n1o.run(t, cond, [], [], has_group=False)
        gpu, stats = pu.run_gpu(t, cond, [], [], filter_only=True, batches=2, filter_stream=stream)
        assert np.array_equal(gpu.selected, ora.selected)


@pytest.mark.parametrize("n", [0, 1, 63, 8192, 8193, 600_001, 3_000_001])
def test_one_pass_filter_ragged_sizes_and_selectivities(n):
    """The one-pass Filter-only kernel (chained scan with decoupled look-back) against the oracle: no rows, one row, sizes
    around a tile (1024 rows), more tiles than one look-back window (64), millions of rows (thousands of tiles in flight);
    conditions nothing passes, everything passes, about half passes; one batch and three (ordinals continue across batches)."""
    t = n1o.synth_table(n, k_cat=7)
    for cond in ["(%s < -1)" % D("price"), "(%s is not missing)" % D("cat"), "(50 < %s)" % D("price")]:
        ora = n1o.run(t, cond, [], [], has_group=False)
        for batches in (1, 3):
            gpu, stats = pu.run_gpu(t, cond, [], [], filter_only=True, batches=batches, device_resident=n > 100_000)
            assert np.array_equal(gpu.selected, ora.selected), (cond, batches)
            assert stats["rows_selected"] == len(ora.selected)


def test_synth_generator_matches_cpu():
    import ctypes as C
    import torch
    n, k = 50_000, 100
    for zipf in (False, True):
        host = n1o.synth_table(n, k_cat=k, zipf=zipf, first_row=12345, total_rows=10_000_000)
        cat = torch.empty(n, dtype=torch.int32, device="cuda")
        pt = torch.empty(n, dtype=torch.uint8, device="cuda")
        pp = torch.empty(n, dtype=torch.int64, device="cuda")
        ut = torch.empty(n, dtype=torch.uint8, device="cuda")
        up = torch.empty(n, dtype=torch.int64, device="cuda")
        rt = torch.empty(n, dtype=torch.uint8, device="cuda")
        rp = torch.empty(n, dtype=torch.int64, device="cuda")
        cdf = torch.from_numpy(n1o.zipf_cdf(k)).cuda() if zipf else None
        spec = _ffi.SynthSpec(0x5EED0001, 12345, n, 10_000_000, k, 1 if zipf else 0, cdf.data_ptr() if zipf else None)
        st = _ffi.lib().n1k_synth_columns(0, None, C.byref(spec), cat.data_ptr(), pt.data_ptr(), pp.data_ptr(),
                                          ut.data_ptr(), up.data_ptr(), rt.data_ptr(), rp.data_ptr())
        assert st == _ffi.OK
        torch.cuda.synchronize()
        c = host.columns
        assert np.array_equal(cat.cpu().numpy().view(np.uint32), c[0].codes)
        assert np.array_equal(pt.cpu().numpy(), c[1].tags)
        assert np.array_equal(pp.cpu().numpy().view(np.uint64), c[1].payload)
        assert np.array_equal(up.cpu().numpy().view(np.uint64), c[2].payload)
        assert np.array_equal(rp.cpu().numpy().view(np.uint64), c[3].payload)


def test_reopen_and_stop():
    t = n1o.synth_table(10_000, k_cat=5)
    cond, keys, aggs = CONFIG2
    pj = query_amd.plan.filter_group_plan(cond, keys, aggs)
    op = query_amd.GpuFilterGroup(pj)
    by = {c.name: c for c in t.columns}
    cols = [by[p] for p in op.column_paths]
    op.process_items(cols, t.dictionary)
    first = op.after_items()
    op.reopen()  # ≙ reopen(): groups dropped, plan + dictionary kept
    op.process_items(cols, t.dictionary)
    second = op.after_items()
    assert sorted(first.keys) == sorted(second.keys)
    # the same groups with the same values (float sums to the tolerance: the merge of the workgroups' tables is atomic)
    a, b = dict(zip(first.keys, first.aggs)), dict(zip(second.keys, second.aggs))
    for k in a:
        assert all(pu.values_match(x, y, float_agg=True) for x, y in zip(a[k], b[k])), (k, a[k], b[k])
    pu.assert_same_groups(second, n1o.run(t, cond, keys, aggs), aggs=aggs)
    op.send_stop()
    with pytest.raises(query_amd.N1kError) as ei:
        op.process_items(cols, t.dictionary)
    assert ei.value.status == _ffi.STOPPED
    op.done()


def test_error_paths_are_reported_not_silent():
    """Run-time conditions outside the device subset surface as status codes (never as wrong groups)."""
    t = n1o.synth_table(50_000, k_cat=2000)
    by = {c.name: c for c in t.columns}
    # 1. group table capacity exceeded -> N1K_OOM (the caller can raise max_groups)
    op = query_amd.GpuFilterGroup(query_amd.plan.filter_group_plan(None, [D("user_id")], ["count(*)"]), max_groups=64)
    op.process_items([by[p] for p in op.column_paths], t.dictionary)
    with pytest.raises(query_amd.N1kError) as ei:
        op.after_items()
    assert ei.value.status == _ffi.OOM
    op.done()
    # 2. a batch with the wrong number of columns -> N1K_INVALID
    op = query_amd.GpuFilterGroup(query_amd.plan.filter_group_plan(None, [D("cat")], ["sum(%s)" % D("price")]))
    with pytest.raises(query_amd.N1kError) as ei:
        op.process_items([by[D("cat")]], t.dictionary)
    assert ei.value.status == _ffi.INVALID
    # 3. a column that changes kind between batches -> N1K_INVALID
    op.process_items([by[p] for p in op.column_paths], t.dictionary)
    wrong = n1o.Column(D("cat"), n1o.COL_TAGGED64, tags=np.full(10, n1o.T_STRING, np.uint8), payload=np.zeros(10, np.uint64))
    with pytest.raises(query_amd.N1kError) as ei:
        op.process_items([wrong if p == D("cat") else _head(by[p], 10) for p in op.column_paths], t.dictionary)
    assert ei.value.status == _ffi.INVALID
    op.done()


def _head(c, n):
    if c.kind == n1o.COL_DICT32:
        return n1o.Column(c.name, c.kind, codes=c.codes[:n])
    return n1o.Column(c.name, c.kind, tags=c.tags[:n], payload=c.payload[:n])


def test_empty_batches_and_zero_rows():
    t = n1o.synth_table(1000, k_cat=5)
    by = {c.name: c for c in t.columns}
    aggs = sorted(["count(*)", "sum(%s)" % D("price")])
    op = query_amd.GpuFilterGroup(query_amd.plan.filter_group_plan("(50 < %s)" % D("price"), [D("cat")], aggs))
    cols = [by[p] for p in op.column_paths]
    op.process_items([_head(c, 0) for c in cols], t.dictionary)  # an empty batch is legal
    op.process_items(cols, t.dictionary)
    op.process_items([_head(c, 0) for c in cols], t.dictionary)
    gpu = op.after_items()
    op.done()
    ora = n1o.run(t, "(50 < %s)" % D("price"), [D("cat")], aggs)
    pu.assert_same_groups(gpu, ora, aggs=aggs)


def _f64(x):
    return np.array([x], np.float64).view(np.uint64)[0]


def _wide_key_table(n, seed=5):
    """A key column holding every scalar class, numbers a packed field cannot hold (non-integral floats, ints
    beyond +-2^58, the int64 extremes, +-Inf) and the pairs that must collide (5 / 5.0, 0 / -0.0, -2^63 / -2^63.0)."""
    rng = np.random.default_rng(seed)
    pool = [(n1o.T_INT, 5), (n1o.T_FLOAT, _f64(5.0)), (n1o.T_INT, 0), (n1o.T_FLOAT, _f64(-0.0)),
            (n1o.T_FLOAT, _f64(2.5)), (n1o.T_FLOAT, _f64(-2.5)), (n1o.T_FLOAT, _f64(1e300)), (n1o.T_FLOAT, _f64(1e-300)),
            (n1o.T_FLOAT, _f64(float("inf"))), (n1o.T_FLOAT, _f64(float("-inf"))),
            (n1o.T_INT, np.uint64(2 ** 63 - 1)), (n1o.T_INT, np.int64(-2 ** 63).view(np.uint64)),
            (n1o.T_FLOAT, _f64(-2.0 ** 63)), (n1o.T_FLOAT, _f64(2.0 ** 63)), (n1o.T_INT, np.uint64(2 ** 60)),
            (n1o.T_INT, np.int64(-2 ** 60).view(np.uint64)), (n1o.T_FLOAT, _f64(2.0 ** 60)),
            (n1o.T_INT, np.uint64(4617878467915022336)),  # the int whose bits spell 5.5
            (n1o.T_FLOAT, _f64(5.5)), (n1o.T_NULL, 0), (n1o.T_MISSING, 0), (n1o.T_TRUE, 0), (n1o.T_FALSE, 0),
            (n1o.T_STRING, 0), (n1o.T_STRING, 1)]
    pick = rng.integers(0, len(pool), n)
    tags = np.array([pool[i][0] for i in pick], np.uint8)
    pay = np.array([int(pool[i][1]) for i in pick], np.uint64)
    # plus a tail of many distinct floats and wide ints
    m = n // 4
    tags[:m] = n1o.T_FLOAT
    pay[:m] = (rng.integers(0, 3000, m) + 0.25).view(np.uint64)
    tags[m:2 * m] = n1o.T_INT
    pay[m:2 * m] = (rng.integers(0, 3000, m).astype(np.int64) * np.int64(-(2 ** 59 + 1))).view(np.uint64)
    vals = rng.integers(-50, 50, n).astype(np.int64)
    cat = rng.integers(0, 7, n).astype(np.uint32)
    return n1o.Table([n1o.Column(D("k"), n1o.COL_TAGGED64, tags=tags, payload=pay),
                      n1o.Column(D("v"), n1o.COL_TAGGED64, tags=np.full(n, n1o.T_INT, np.uint8), payload=vals.view(np.uint64)),
                      n1o.Column(D("cat"), n1o.COL_DICT32, codes=cat)],
                     [b"s%d" % i for i in range(7)])


@pytest.mark.parametrize("keys", [[D("k")], [D("cat"), D("k")], [D("k"), D("k")]])
@pytest.mark.parametrize("opts", [{}, {"spec": 0}, {"fast": 0, "spec": 0}, {"jit": 2}], ids=["auto", "fast", "interp", "jit"])
def test_wide_key_values_group_like_the_reference(keys, opts):
    """execution/group_util.go:18-35 groups on the canonical JSON of the key values: floats, huge ints and the
    int/float pairs that print alike.  On the device such numbers are keyed by their value-table code."""
    t = _wide_key_table(40_000)
    aggs = sorted(["count(*)", "sum(%s)" % D("v"), "min(%s)" % D("v")])
    ora = n1o.run(t, None, keys, aggs, threads=2)
    gpu, stats = pu.run_gpu(t, None, keys, aggs, batches=3, **opts)
    pu.assert_same_groups(gpu, ora, aggs=aggs)
    assert stats["wide_key_values"] >= 5000


@pytest.mark.parametrize("opts", [{}, {"fast": 0, "spec": 0}, {"jit": 2}, {"agg_mode": 4}], ids=["auto", "interp", "jit", "partitioned"])
def test_group_keys_are_their_json_text(opts):
    """execution/group_util.go:18-35 groups on the MARSHALLED key (value/float.go:31-48), with two consequences only computed or
    column-fed floats can reach: an integral float from 2^53 up prints its shortest digits followed by zeros — float 2^60 is
    "1152921504606847000" — and is one group with the INT of that text; NaN / +Inf / -Inf print as the JSON strings "NaN" /
    "+Infinity" / "-Infinity" and are one group with those strings.  Floats beyond int64 and ordinary floats keep their own."""
    f60 = float(2 ** 60)
    rows = [(n1o.T_FLOAT, f60), (n1o.T_INT, 1152921504606847000), (n1o.T_INT, 2 ** 60),      # two groups: the text, and the exact int
            (n1o.T_FLOAT, -float(2 ** 62)), (n1o.T_INT, -4611686018427388000),                 # one group
            (n1o.T_FLOAT, float(2 ** 53)), (n1o.T_INT, 2 ** 53), (n1o.T_INT, 2 ** 53 + 1),    # 9007199254740992: float and int alike
            (n1o.T_FLOAT, float("nan")), (n1o.T_STRING, b"NaN"), (n1o.T_FLOAT, float("inf")), (n1o.T_STRING, b"+Infinity"),
            (n1o.T_FLOAT, float("-inf")), (n1o.T_STRING, b"-Infinity"), (n1o.T_STRING, b"Infinity"),
            (n1o.T_FLOAT, float(2 ** 63)), (n1o.T_FLOAT, 1e300), (n1o.T_FLOAT, 2.5), (n1o.T_FLOAT, 7.0), (n1o.T_INT, 7)]
    rows = rows * 3
    dictionary = [b"NaN", b"+Infinity", b"-Infinity", b"Infinity"]
    n = len(rows)
    tags = np.array([t for t, _ in rows], np.uint8)
    pay = np.zeros(n, np.uint64)
    for i, (t, v) in enumerate(rows):
        pay[i] = (np.int64(v).view(np.uint64) if t == n1o.T_INT else np.float64(v).view(np.uint64) if t == n1o.T_FLOAT
                  else np.uint64(dictionary.index(v)))
    vals = np.arange(n, dtype=np.int64)
    t = n1o.Table([n1o.Column(D("k"), n1o.COL_TAGGED64, tags=tags, payload=pay),
                   n1o.Column(D("v"), n1o.COL_TAGGED64, tags=np.full(n, n1o.T_INT, np.uint8), payload=vals.view(np.uint64))], dictionary)
    aggs = sorted(["count(*)", "sum(%s)" % D("v")])
    ora = n1o.run(t, None, [D("k")], aggs, threads=1)
    assert len(ora.keys) == 13  # 20 distinct values, 13 distinct texts
    gpu, _ = pu.run_gpu(t, None, [D("k")], aggs, batches=2, **opts)
    pu.assert_same_groups(gpu, ora, aggs=aggs)


def test_wide_key_value_table_overflow_is_reported():
    t = _wide_key_table(40_000)
    with pytest.raises(query_amd.N1kError) as ei:
        pu.run_gpu(t, None, [D("k")], ["count(*)"], wide_values=64)
    assert ei.value.status == _ffi.UNSUPPORTED_DATA and "wide_values" in ei.value.message


def test_wide_key_values_survive_reopen():
    t = _wide_key_table(20_000)
    pj = query_amd.plan.filter_group_plan(None, [D("k")], ["count(*)"])
    op = query_amd.GpuFilterGroup(pj)
    by = {c.name: c for c in t.columns}
    ora = n1o.run(t, None, [D("k")], ["count(*)"])
    for _ in range(2):
        op.process_items([by[p] for p in op.column_paths], t.dictionary)
        pu.assert_same_groups(op.after_items(), ora)
        op.reopen()
    op.done()


def _one_group(values, aggs_of_v):
    """values: [(tag, python value)] of one column `v`, all rows in one group; the oracle runs one thread, rows in order."""
    n = len(values)
    tags = np.array([t for t, _ in values], np.uint8)
    pay = np.zeros(n, np.uint64)
    for i, (t, v) in enumerate(values):
        pay[i] = np.int64(v).view(np.uint64) if t == n1o.T_INT else (np.float64(v).view(np.uint64) if t == n1o.T_FLOAT else 0)
    t = n1o.Table([n1o.Column(D("v"), n1o.COL_TAGGED64, tags=tags, payload=pay)], [])
    aggs = sorted(a % D("v") for a in aggs_of_v)
    ora = n1o.run(t, None, [], aggs, threads=1)
    gpu, _ = pu.run_gpu(t, None, [], aggs)
    return dict(zip(aggs, gpu.aggs[0])), dict(zip(aggs, ora.aggs[0]))


def test_corners_the_comparison_relaxes_checked_where_the_reference_is_well_defined():
    """tests/parity_util.values_match relaxes three corners for the random plans (tie_ok, the 2^53 image rule, folded
    AVG).  Each has inputs on which the reference's answer does not depend on arrival order; there the device must be
    bit-exact, tag included — fixed row order, one oracle thread, no tolerance."""
    I, F = n1o.T_INT, n1o.T_FLOAT
    exact = lambda g, o: g == o
    # (1) MIN / MAX over an int and the float equal to it: Collate ties, the first to arrive stays (agg_min.go:83-94).
    #     With the INT first the answer is the INT on both sides, whatever follows.
    g, o = _one_group([(I, 7), (F, 7.0), (I, 9), (F, 9.0), (F, 8.5)], ["min(%s)", "max(%s)"])
    assert all(exact(g[a], o[a]) for a in g), (g, o)
    assert g["min(%s)" % D("v")] == (I, 7) and g["max(%s)" % D("v")] == (I, 9)
    #     With the FLOAT first the reference keeps the float; the device reports the int.  Both marshal to the same JSON
    #     ("7": value/float.go:31-48 prints integral floats without a fraction) — the one documented difference.
    g, o = _one_group([(F, 7.0), (I, 7), (F, 6.5 + 0.5)], ["min(%s)"])
    assert o["min(%s)" % D("v")] == (F, 7.0) and g["min(%s)" % D("v")] == (I, 7)
    assert gu.canonical_json(o["min(%s)" % D("v")][1]) == gu.canonical_json(g["min(%s)" % D("v")][1]) == "7"
    # (2) ints beyond 2^53 whose float64 images coincide: two ints compare exactly (value/integer.go:100-118), so
    #     without a float among them MIN / MAX are exact
    big = 2 ** 53
    g, o = _one_group([(I, big + 3), (I, big + 1), (I, big + 2), (I, -(big + 1)), (I, -(big + 2))], ["min(%s)", "max(%s)", "count(%s)"])
    assert all(exact(g[a], o[a]) for a in g), (g, o)
    assert g["max(%s)" % D("v")] == (I, big + 3) and g["min(%s)" % D("v")] == (I, -(big + 2))
    # (3) AVG folds float64(sum) / float64(count) to INT when integral (agg_avg.go:136-157): with a sum and a quotient
    #     that float64 holds exactly, exactly that INT
    g, o = _one_group([(I, 2 ** 52), (I, 2 ** 52 + 2)], ["avg(%s)", "sum(%s)", "count(%s)"])
    assert all(exact(g[a], o[a]) for a in g), (g, o)
    assert g["avg(%s)" % D("v")] == (I, 2 ** 52 + 1)
    # (4) what the random generator steers around — signed divisors, signed x huge products — value by value
    exprs = ["(%s / -4)", "(%s * -2305843009213693952)", "(%s %% -7)", "idiv(%s, -3)", "imod(%s, -3)", "((-%s) - 9223372036854775807)"]
    vals = [(I, 10), (I, -10), (I, 7), (F, 2.5), (I, 4), (I, -(2 ** 62)), (I, 2 ** 62)]
    n = len(vals)
    tags = np.array([t for t, _ in vals], np.uint8)
    pay = np.array([np.int64(v).view(np.uint64) if t == I else np.float64(v).view(np.uint64) for t, v in vals], np.uint64)
    t = n1o.Table([n1o.Column(D("id"), n1o.COL_TAGGED64, tags=np.full(n, I, np.uint8), payload=np.arange(n, dtype=np.uint64)),
                   n1o.Column(D("v"), n1o.COL_TAGGED64, tags=tags, payload=pay)], [])
    for e in exprs:
        aggs = ["max(%s)" % (e % D("v"))]  # one row per group: MAX is the value of the expression
        ora = n1o.run(t, None, [D("id")], aggs, threads=1)
        gpu, _ = pu.run_gpu(t, None, [D("id")], aggs)
        assert dict(zip(gpu.keys, gpu.aggs)) == dict(zip(ora.keys, ora.aggs)), e


def test_string_min_max_when_the_dictionary_grows_between_batches():
    """MIN / MAX over strings compare by bytewise rank (value/string.go:116-130).  The ranks belong to the dictionary
    of the moment: a batch that brings strings sorting BEFORE the winners kept so far must still be compared in the
    new order (the kept winners are re-stamped).  Batch 1 knows {m, z}; batch 2 adds {a, b, zz}."""
    keys, aggs = [D("g")], sorted(["min(%s)" % D("s"), "max(%s)" % D("s"), "count(*)"])

    def batch(groups, strings, dictionary):
        g = np.array(groups, dtype=np.uint64)
        codes = np.array([dictionary.index(x) for x in strings], dtype=np.uint64)
        n = len(groups)
        return n1o.Table([n1o.Column(D("g"), n1o.COL_TAGGED64, tags=np.full(n, n1o.T_INT, np.uint8), payload=g),
                          n1o.Column(D("s"), n1o.COL_TAGGED64, tags=np.full(n, n1o.T_STRING, np.uint8), payload=codes)], dictionary)

    d1, d2 = [b"m", b"z"], [b"a", b"b", b"m", b"zz"]
    b1 = batch([0, 0, 1, 1, 2], [b"z", b"m", b"m", b"m", b"z"], d1)
    b2 = batch([0, 1, 1, 2, 3], [b"m", b"a", b"zz", b"b", b"a"], d2)
    union = [b"m", b"z", b"a", b"b", b"zz"]
    whole = batch([0, 0, 1, 1, 2, 0, 1, 1, 2, 3], [b"z", b"m", b"m", b"m", b"z", b"m", b"a", b"zz", b"b", b"a"], union)
    ora = n1o.run(whole, None, keys, aggs)
    for json_docs in (False, True):
        op = query_amd.GpuFilterGroup(query_amd.plan.filter_group_plan(None, keys, aggs))
        if json_docs:  # the streaming path of the finding: every n1k_push_json interns the batch's new strings
            op.process_json([b'{"g": %d, "s": "%s"}' % (g, s) for g, s in zip([0, 0, 1, 1, 2], [b"z", b"m", b"m", b"m", b"z"])])
            op.process_json([b'{"g": %d, "s": "%s"}' % (g, s) for g, s in zip([0, 1, 1, 2, 3], [b"m", b"a", b"zz", b"b", b"a"])])
        else:
            for b in (b1, b2):
                op.process_items([{c.name: c for c in b.columns}[p] for p in op.column_paths], b.dictionary)
        rows = op.after_items()
        op.done()
        pu.assert_same_groups(rows, ora, aggs=aggs)
        got = {k[0][1]: a for k, a in zip(rows.keys, rows.aggs)}
        mx, mn = aggs.index("max(%s)" % D("s")), aggs.index("min(%s)" % D("s"))
        assert got[0][mn][1] == b"m" and got[0][mx][1] == b"z" and got[1][mn][1] == b"a" and got[1][mx][1] == b"zz"
        assert got[2][mn][1] == b"b" and got[2][mx][1] == b"z"


def _distinct_table(n, nvals, ngroups, seed=3, wide_share=0.1):
    """(g, v): v mostly small ints (one-word members), plus floats / huge ints / strings (two-word pairs)."""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, ngroups, n).astype(np.uint64)
    tags = np.full(n, n1o.T_INT, np.uint8)
    pay = (rng.integers(0, nvals, n) - nvals // 3).astype(np.int64).view(np.uint64).copy()
    w = rng.random(n) < wide_share
    kind = rng.integers(0, 4, n)
    fl = w & (kind == 0)
    tags[fl] = n1o.T_FLOAT
    pay[fl] = (rng.integers(0, nvals, int(fl.sum())) + 0.5).view(np.uint64)
    big = w & (kind == 1)
    pay[big] = (rng.integers(0, 50, int(big.sum())).astype(np.int64) * np.int64(2 ** 55)).view(np.uint64)
    st = w & (kind == 2)
    tags[st] = n1o.T_STRING
    pay[st] = rng.integers(0, 3, int(st.sum())).astype(np.uint64)
    nul = w & (kind == 3)
    tags[nul] = n1o.T_NULL
    pay[nul] = 0
    return n1o.Table([n1o.Column(D("g"), n1o.COL_TAGGED64, tags=np.full(n, n1o.T_INT, np.uint8), payload=g),
                      n1o.Column(D("v"), n1o.COL_TAGGED64, tags=tags, payload=pay)], [b"x", b"y", b"z"])


@pytest.mark.parametrize("n,nvals,ngroups,opts,path", [
    (3_000, 50, 4, {}, 3),                                  # one LDS set for the whole log (no partition pass)
    (120_000, 40_000, 7, {}, 3),                            # one partition pass
    (900_000, 700_000, 50, {}, 3),                          # two partition passes
    (400_000, 16, 5, {}, 3),                                # few members, each logged many times (filter cache, hot bins)
    (300_000, 200_000, 9, {"distinct_set_slots": 64, "distinct_levels": 1}, 7),   # bins overflow their LDS sets: global one-word set
    (200_000, 90_000, 6, {"distinct_words": 0}, 1),         # everything through the two-word pair sets
], ids=["direct", "one-pass", "two-pass", "duplicates", "overflow-fallback", "pairs-only"])
def test_count_distinct_paths(n, nvals, ngroups, opts, path):
    """COUNT(DISTINCT) through every way the sets are built (value.Set semantics, value/set.go:22-110)."""
    t = _distinct_table(n, nvals, ngroups)
    aggs = sorted(["count(distinct %s)" % D("v"), "count(%s)" % D("v")])
    ora = n1o.run(t, None, [D("g")], aggs, threads=2)
    gpu, stats = pu.run_gpu(t, None, [D("g")], aggs, batches=2, **opts)
    pu.assert_same_groups(gpu, ora, aggs=aggs)
    assert stats["distinct_path"] == path


@pytest.mark.parametrize("n,nvals,ngroups,opts,path", [
    (3_000, 50, 4, {}, 3),                                   # a few words per hash region: LDS sets straight over the regions
    (120_000, 40_000, 7, {}, 3),
    (900_000, 700_000, 50, {}, 3),                           # second partition pass into bins of fixed capacity
    (900_000, 700_000, 50, {"distinct_levels": 2, "distinct_set_slots": 1024}, 3),
    (400_000, 16, 5, {}, 3),                                 # few members, each logged many times (workgroup caches)
    (400_000, 16, 5, {"distinct_region_cap": 64}, 3),        # regions overflow into the plain log: gathered, exact path
    (300_000, 200_000, 9, {"distinct_region_cap": 300}, 3),
    (300_000, 200_000, 9, {"distinct_set_slots": 64, "distinct_levels": 1}, 7),  # LDS sets overflow: global one-word set
    (3_000, 50, 4, {"distinct_levels": 0}, 3),              # forced depth 0: gathered, one set
], ids=["tiny", "regions", "two-pass", "two-pass-forced", "duplicates", "spill-dups", "spill", "set-overflow", "gathered"])
@pytest.mark.parametrize("batches", [1, 3])
def test_count_distinct_in_the_specialised_scan(n, nvals, ngroups, opts, path, batches):
    """COUNT(DISTINCT) logged by the plan-specialised scan (built at run time for this shape: integer key, open-addressed
    LDS table): member words go straight into the 256 hash regions (first partition pass fused into the scan), the
    rest of value.Set's semantics (value/set.go:22-110) as in the interpreter path; two-word members take the pair log."""
    t = _distinct_table(n, nvals, ngroups)
    aggs = sorted(["count(distinct %s)" % D("v"), "count(%s)" % D("v")])
    ora = n1o.run(t, None, [D("g")], aggs, threads=2)
    gpu, stats = pu.run_gpu(t, None, [D("g")], aggs, batches=batches, jit=2, **opts)
    assert stats["spec_kernel"] == 2
    pu.assert_same_groups(gpu, ora, aggs=aggs)
    assert stats["distinct_path"] == path


def test_count_distinct_regions_grow_and_mix_with_the_interpreter():
    """The hash regions grow with the rows pushed (their words move to the wider layout), and a query whose batches ran
    through both kernels (regions + plain word log) is finished through the exact path."""
    t = n1o.synth_table(600_000, k_cat=50, total_rows=3_000_000)
    cond, keys, aggs = CONFIG3
    ora = n1o.run(t, cond, keys, aggs, threads=2)
    pj = query_amd.plan.filter_group_plan(cond, keys, aggs)
    by = {c.name: c for c in t.columns}
    cuts = [0, 1_000, 5_000, 40_000, 300_000, 600_000]  # growing batches: the regions are re-laid out several times
    for mixed in (False, True):
        op = query_amd.GpuFilterGroup(pj)
        for i in range(len(cuts) - 1):
            part = t.slice(cuts[i], cuts[i + 1])
            if mixed:
                op.set_option("spec", i % 2)  # every other batch through the interpreter kernel
            op.process_items([{c.name: c for c in part.columns}[p] for p in op.column_paths], t.dictionary)
        rows = op.after_items()
        assert op.stats()["distinct_path"] & 2
        op.done()
        pu.assert_same_groups(rows, ora, aggs=aggs)


def test_config3_shape_runs_the_prebuilt_specialised_kernel():
    t = n1o.synth_table(300_000, k_cat=1000, total_rows=3_000_000)
    cond, keys, aggs = CONFIG3
    ora = n1o.run(t, cond, keys, aggs, threads=2)
    gpu, stats = pu.run_gpu(t, cond, keys, aggs, device_resident=True)
    assert stats["spec_kernel"] == 1 and stats["distinct_path"] == 2
    pu.assert_same_groups(gpu, ora, aggs=aggs)


def test_count_distinct_words_refinish_and_reopen():
    """The word log survives n1k_finish (more batches may follow) and is dropped by reopen."""
    t = _distinct_table(150_000, 60_000, 5)
    aggs = ["count(distinct %s)" % D("v")]
    pj = query_amd.plan.filter_group_plan(None, [D("g")], aggs)
    op = query_amd.GpuFilterGroup(pj)
    by = {c.name: c for c in t.columns}
    half = t.slice(0, 75_000), t.slice(75_000, 150_000)
    for _ in range(2):
        op.process_items([{c.name: c for c in half[0].columns}[p] for p in op.column_paths], t.dictionary)
        pu.assert_same_groups(op.after_items(), n1o.run(half[0], None, [D("g")], aggs))
        op.process_items([{c.name: c for c in half[1].columns}[p] for p in op.column_paths], t.dictionary)
        pu.assert_same_groups(op.after_items(), n1o.run(t, None, [D("g")], aggs))
        op.reopen()
    op.done()


def test_count_distinct_dictionary_key_and_strings():
    """config 3's shape with a dictionary-coded key; string operands are one-word members too when codes are small."""
    t = n1o.synth_table(150_000, k_cat=300, total_rows=40_000)
    aggs = sorted(["count(distinct %s)" % D("user_id"), "count(distinct %s)" % D("cat"), "count(distinct %s)" % D("price")])
    ora = n1o.run(t, None, [D("cat")], aggs, threads=2)
    gpu, stats = pu.run_gpu(t, None, [D("cat")], aggs, batches=2)
    pu.assert_same_groups(gpu, ora, aggs=aggs)
    assert stats["distinct_path"] & 2
    ora = n1o.run(t, None, [], aggs, threads=2)
    gpu, stats = pu.run_gpu(t, None, [], aggs)
    pu.assert_same_groups(gpu, ora, aggs=aggs)


ORDER_KEYS = [D("cat"), D("region_id")]
ORDER_AGGS = sorted(["sum(%s)" % D("price"), "count(*)", "min(%s)" % D("price")])


@pytest.mark.parametrize("order,limit,offset", [
    ([("sum(%s)" % D("price"), True)], 100, None),                                    # config 5's tail
    ([("count(*)", True), (D("cat"), False), (D("region_id"), False)], None, None),   # a total order: exact sequence
    ([(D("cat"), False), (D("region_id"), True)], 20, 5),
    ([("min(%s)" % D("price"), False), ("count(*)", True)], 7, 3),                    # mixed-type term (numbers, "n/a", NULL)
    (None, 10, 2),                                                                    # Limit / Offset without Order
    ([("sum(%s)" % D("price"), False)], 0, None),
    ([("sum(%s)" % D("price"), True)], 100000, 10),                                   # limit beyond the groups
], ids=["sum-desc-limit", "total-order", "keys-offset-limit", "mixed-types", "limit-only", "limit-0", "limit-beyond"])
def test_order_by_limit_over_the_groups(order, limit, offset):
    """BASELINE config 5's tail: ORDER BY <aggregate | key> [DESC] OFFSET o LIMIT k over the final groups
    (execution/order.go:121-169, order_limit.go, offset.go, limit.go) inside the same handle."""
    t = n1o.synth_table(90_000, k_cat=60)
    ora = n1o.run(t, None, ORDER_KEYS, ORDER_AGGS, threads=2)
    gpu, _ = pu.run_gpu(t, None, ORDER_KEYS, ORDER_AGGS, batches=2, order=order, limit=limit, offset=offset)
    pu.assert_ordered_groups(gpu, ora, ORDER_KEYS, ORDER_AGGS, order, limit, offset)


def test_order_by_rejects_terms_outside_the_groups():
    pj = query_amd.plan.filter_group_plan(None, ORDER_KEYS, ORDER_AGGS, order=[(D("price"), False)])
    with pytest.raises(query_amd.N1kError) as ei:
        query_amd.GpuFilterGroup(pj)
    assert ei.value.status == _ffi.UNSUPPORTED


@pytest.mark.parametrize("order,limit,offset", [
    ([("sum(%s)" % D("price"), True)], 100, None),
    ([("count(*)", False), (D("cat"), True)], 30, 10),           # many ties on the first term: all of them are candidates
    ([(D("cat"), True), (D("region_id"), False)], 50, 0),        # string term: images through the bytewise rank
    ([("min(%s)" % D("price"), True)], 5, 1),                    # strings ("n/a") sort above numbers
])
def test_device_topk_filter_feeds_the_exact_order(order, limit, offset):
    """execution/order_limit.go: with many groups only the candidates for the first offset+limit rows (order image of
    the first term <= the selected threshold) are copied to the host; the final order is the exact collation."""
    t = n1o.synth_table(150_000, k_cat=120)
    ora = n1o.run(t, None, ORDER_KEYS, ORDER_AGGS, threads=2)
    gpu, stats = pu.run_gpu(t, None, ORDER_KEYS, ORDER_AGGS, order=order, limit=limit, offset=offset, topk_min_groups=1)
    pu.assert_ordered_groups(gpu, ora, ORDER_KEYS, ORDER_AGGS, order, limit, offset)
    assert 0 < stats["topk_candidates"] < len(ora.keys)


@pytest.mark.parametrize("lean", [1, 0], ids=["order-values-first", "every-row"])
@pytest.mark.parametrize("order,limit,offset", [
    ([("sum(%s)" % D("price"), True)], 100, None),
    ([(D("cat"), True), (D("region_id"), False)], 50, 0),        # a key term: its value is unpacked from the packed key
    ([("count(*)", False), (D("cat"), True)], 30, 10),
])
def test_topk_over_the_partitioned_paths_kept_region(order, limit, offset, lean):
    """ORDER BY ... LIMIT over groups that stayed in the partitioned path's compact region (one batch, nothing else in
    the handle): FinalGroup first writes only the first term's value of every group, the top-k filter picks the
    candidates, and only their rows are finalised (lean_topk, default) — same rows as finalising every group."""
    t = n1o.synth_table(150_000, k_cat=120)
    ora = n1o.run(t, None, ORDER_KEYS, ORDER_AGGS, threads=2)
    gpu, stats = pu.run_gpu(t, None, ORDER_KEYS, ORDER_AGGS, order=order, limit=limit, offset=offset, topk_min_groups=1,
                            agg_mode=4, lean_topk=lean)
    pu.assert_ordered_groups(gpu, ora, ORDER_KEYS, ORDER_AGGS, order, limit, offset)
    assert stats["agg_mode"] == 4 and 0 < stats["topk_candidates"] < len(ora.keys)


@pytest.mark.parametrize("sample", [1, 0], ids=["sampled-threshold", "radix-select"])
@pytest.mark.parametrize("order,limit,offset", [
    ([("sum(%s)" % D("price"), True)], 100, None),               # config 5's tail
    ([("count(*)", True), (D("cat"), False)], 40, 5),             # few distinct first-term values: a flood of ties below any threshold
    ([("sum(%s)" % D("price"), False), (D("cat"), True)], 2000, 0),  # a limit the sample's rank has to follow
])
def test_topk_threshold_from_a_sample(order, limit, offset, sample):
    """Hundreds of thousands of groups: the device top-k filter takes its threshold from a sample of the groups' order images
    (topk_sample_kernel) and the host checks that at least offset + limit candidates came out (else the exact radix select
    runs); either way the rows are the oracle's, in the reference's order — against the exact radix select as well."""
    t = n1o.synth_table(400_000, k_cat=50_000)
    keys, aggs = [D("cat"), D("region_id")], sorted(["sum(%s)" % D("price"), "count(*)"])
    ora = n1o.run(t, None, keys, aggs, threads=4)
    assert len(ora.keys) > 4 * 16384  # (enough groups for the sampled path)
    gpu, stats = pu.run_gpu(t, None, keys, aggs, order=order, limit=limit, offset=offset, topk_sample=sample, device_resident=True)
    pu.assert_ordered_groups(gpu, ora, keys, aggs, order, limit, offset)
    assert (limit + (offset or 0)) <= stats["topk_candidates"] < len(ora.keys)


PART_AGGS = sorted(["sum(%s)" % D("price"), "count(*)", "min(%s)" % D("price"), "avg(%s)" % D("price"),
                    "max(%s)" % D("region_id"), "countn(%s)" % D("price")])


@pytest.mark.parametrize("keys,cond,levels", [
    ([D("user_id")], None, -1),                                     # ~18 k groups of 200 k rows
    ([D("user_id")], "(50 < %s)" % D("price"), 0),                  # no partition pass: one bin, LDS overflow -> global rows
    ([D("cat"), D("region_id")], None, 1),
    ([D("user_id"), D("region_id")], "(%s is valued)" % D("price"), 2),  # two passes, 65 536 bins
    ([D("price")], None, 1),                                        # float keys: wide-value codes inside the records
], ids=["auto-levels", "no-pass", "one-pass", "two-pass", "float-keys"])
def test_partitioned_high_cardinality_group_by(keys, cond, levels):
    """agg_mode 4: rows -> records -> radix partition on the key hash -> per-bin LDS tables; same groups and
    aggregates as the reference whatever the number of passes (execution/group_initial.go, group_intermediate.go)."""
    t = n1o.synth_table(200_000, k_cat=700, total_rows=200_000)
    ora = n1o.run(t, cond, keys, PART_AGGS, threads=2)
    gpu, stats = pu.run_gpu(t, cond, keys, PART_AGGS, batches=2, agg_mode=4, partition_levels=levels)
    pu.assert_same_groups(gpu, ora, aggs=PART_AGGS)
    assert stats["agg_mode"] == 4 and stats["rows_selected"] == ora.rows_passed


def _records_table(n, seed=7):
    """Many groups (an int key), ONE operand column holding every kind of value the per-bin tables meet: small ints, ints beyond
    2^40 (they leave the LDS sum on their own), floats, NULL / MISSING, a boolean and a string now and then."""
    rng = np.random.default_rng(seed)
    key = rng.integers(0, n // 6, n).astype(np.int64)
    kind = rng.integers(0, 100, n)
    tags = np.full(n, n1o.T_INT, np.uint8)
    pay = rng.integers(-1000, 1000, n).astype(np.int64).view(np.uint64).copy()
    big = kind < 3
    pay[big] = (rng.integers(1, 1 << 20, big.sum()).astype(np.int64) << 41).view(np.uint64)
    fl = (kind >= 3) & (kind < 40)
    tags[fl] = n1o.T_FLOAT
    pay[fl] = np.round(rng.uniform(-50, 50, fl.sum()), 3).view(np.uint64)
    tags[(kind >= 40) & (kind < 45)] = n1o.T_NULL
    tags[(kind >= 45) & (kind < 50)] = n1o.T_MISSING
    tags[kind == 50] = n1o.T_TRUE
    st = kind == 51
    tags[st] = n1o.T_STRING
    pay[st] = rng.integers(0, 3, st.sum()).astype(np.uint64)
    pay[(tags == n1o.T_NULL) | (tags == n1o.T_MISSING) | (tags == n1o.T_TRUE)] = 0
    return n1o.Table([n1o.Column(D("k"), n1o.COL_TAGGED64, tags=np.full(n, n1o.T_INT, np.uint8), payload=key.view(np.uint64)),
                      n1o.Column(D("v"), n1o.COL_TAGGED64, tags=tags, payload=pay)], [b"a", b"b", b"c"])


@pytest.mark.parametrize("aggs", [["sum(%s)"], ["avg(%s)"], ["count(%s)"], ["countn(%s)"], ["min(%s)"], ["max(%s)"], ["count(*)"],
                                  ["count(*)", "max(%s)", "sum(%s)"]],
                         ids=["sum", "avg", "count", "countn", "min", "max", "count-star", "three-of-them"])
@pytest.mark.parametrize("opts", [{}, {"agg_spec": 0}, {"rec_slots": 64}], ids=["specialised", "generic", "tiny-tables"])
def test_per_bin_tables_over_16_byte_records(aggs, opts):
    """agg_bins16_kernel (the partitioned GROUP BY over (key, operand) records): every aggregate kind as the plan's ONE aggregate
    (fixed at compile time) and through the generic form, several at once, operands of every kind — ints beyond 2^40 leave the
    narrow LDS sum as partial groups of their own — and per-bin tables too small for their bins (the rest leaves the same way)."""
    aggs = sorted(a % D("v") if "%s" in a else a for a in aggs)
    t = _records_table(120_000)
    ora = n1o.run(t, None, [D("k")], aggs, threads=2)
    gpu, stats = pu.run_gpu(t, None, [D("k")], aggs, batches=2, agg_mode=4, jit=2, **opts)
    pu.assert_same_groups(gpu, ora, aggs=aggs)
    assert stats["agg_mode"] == 4


def test_partitioned_path_is_chosen_from_the_data():
    """AUTO: the first rows of a large batch run through the scan kernels; many new groups there send the rest of
    the batch through the partitioned path, few keep the scan kernels."""
    t = n1o.synth_table(300_000, k_cat=50, total_rows=3_000_000)
    aggs = sorted(["sum(%s)" % D("price"), "count(*)"])
    opts = dict(partition_min_rows=100_000, partition_probe_rows=50_000, partition_min_groups=10_000)
    ora = n1o.run(t, None, [D("user_id")], aggs, threads=2)          # ~190 k groups
    gpu, stats = pu.run_gpu(t, None, [D("user_id")], aggs, **opts)
    pu.assert_same_groups(gpu, ora, aggs=aggs)
    assert stats["agg_mode"] == 4
    ora = n1o.run(t, None, [D("region_id")], aggs, threads=2)        # 64 groups
    gpu, stats = pu.run_gpu(t, None, [D("region_id")], aggs, **opts)
    pu.assert_same_groups(gpu, ora, aggs=aggs)
    assert stats["agg_mode"] != 4


def _docs_of(table, n):
    """The synthetic table as raw JSON documents ({"id","cat","price","user_id","region_id","pad"})."""
    import json
    by = {c.name: c for c in table.columns}
    docs = []
    for i in range(n):
        d = {"id": "d%d" % i, "pad": "x" * (i % 40)}
        c = int(by[D("cat")].codes[i])
        if c == 0xFFFFFFFE:
            d["cat"] = None
        elif c != 0xFFFFFFFF:
            d["cat"] = table.dictionary[c].decode()
        for name in ("price", "user_id", "region_id"):
            col = by[D(name)]
            t, p = int(col.tags[i]), col.payload[i]
            if t == n1o.T_INT:
                d[name] = int(np.uint64(p).astype(np.int64))
            elif t == n1o.T_FLOAT:
                d[name] = float(np.array([p], np.uint64).view(np.float64)[0])
            elif t == n1o.T_STRING:
                d[name] = table.dictionary[int(p)].decode()
            elif t == n1o.T_NULL:
                d[name] = None
        docs.append(json.dumps(d).encode())
    return docs


@pytest.mark.parametrize("cond,keys,aggs", [CONFIG2, CONFIG3,
                                            (None, [D("cat"), D("region_id")], ["sum(%s)" % D("price")])],
                         ids=["config2", "config3", "config5-keys"])
def test_raw_json_documents_end_to_end(cond, keys, aggs):
    """n1k_push_json: documents in, groups out — the same answers as the oracle over the columns the documents were
    made from (leaf access + typing: value/parsed.go:159-207, value/value.go:367-430)."""
    n = 30_000
    t = n1o.synth_table(n, k_cat=40, total_rows=20_000)
    docs = _docs_of(t, n)
    ora = n1o.run(t, cond, keys, aggs, threads=2)
    op = query_amd.GpuFilterGroup(query_amd.plan.filter_group_plan(cond, keys, aggs))
    op.process_json(docs[:n // 2])
    op.process_json(docs[n // 2:])
    gpu = op.after_items()
    op.done()
    pu.assert_same_groups(gpu, ora, aggs=aggs)


def _tricky_docs(n, rng):
    """Documents that walk the device extractor's branches: nested wanted paths, first-field-wins duplicates, escapes in wanted
    strings and in names, array / object values, numbers at the edges of its exact conversions, documents that are not
    objects, one larger than a wave's LDS share, whitespace everywhere."""
    import json
    docs = []
    strs = ["alpha", "beta", "g\"q", "tab\there", "\u00e9t\u00e9", "", "x" * 300, "NaN"]
    nums = ["0", "-0", "7", "-12", "123456789012345678", "1234567890123456789", "9223372036854775807", "9223372036854775808",
            "1.5", "-2.25", "12.34", "0.1", "1e3", "1E3", "2.5e-3", "1e22", "1e23", "123456789012345.6", "1234567890123456.7",
            "5.0", "-0.0", "1e400", "4.9e-324", "0.30000000000000004", "100e-2", "3.0e0"]
    for i in range(n):
        k = int(rng.integers(0, 12))
        s = json.dumps(strs[int(rng.integers(0, len(strs)))]) if k != 0 else None
        num = nums[int(rng.integers(0, len(nums)))]
        inner = '{"y": %s, "z": [1, {"y": 5}]}' % nums[int(rng.integers(0, len(nums)))]
        if k == 1:
            doc = '{"pad": "p", "s": %s, "n": %s, "x": %s, "s": "second", "n": 99}' % (s, num, inner)  # duplicates: the first counts
        elif k == 2:
            doc = ' {\n\t"n" :%s ,\r\n "x" : %s , "s":%s } ' % (num, inner, s)
        elif k == 3:
            doc = '{"s": %s, "n": [1, 2, {"a": "b"}], "x": {"y": {"deep": true}}}' % s  # array / object values of wanted paths
        elif k == 4:
            doc = '{"\\u0073": "escaped name", "s": %s, "n": %s, "x": 5}' % (s, num)  # "x" is a scalar: x.y is MISSING
        elif k == 5:
            doc = '{"s": %s, "n": %s, "big": "%s", "x": %s}' % (s, num, "b" * 20000, inner)  # larger than 16 KB
        elif k == 6:
            doc = ['42', '[1, 2, 3]', '"just a string"', 'null', '{}'][int(rng.integers(0, 5))]
        elif k == 7:
            doc = '{"n": true, "s": null, "x": {"y": false}}'
        elif k == 8:
            doc = '{"skip": {"s": "inner s does not count", "a": [[], {}, [{"q": "\\\\"}]]}, "s": %s, "n": %s}' % (s, num)
        else:
            doc = '{"id": "d%d", "s": %s, "n": %s, "x": %s, "pad": "%s"}' % (i, s if s else '"none"', num, inner, "x" * int(rng.integers(0, 50)))
        docs.append(doc.encode())
    return docs


def _groups_from_json(docs, device, aggs, keys, min_docs=1):
    op = query_amd.GpuFilterGroup(query_amd.plan.filter_group_plan(None, keys, aggs), json_device=device, json_device_min_docs=min_docs,
                                 json_device_left_pct=100)
    half = len(docs) // 2
    op.process_json(docs[:half])
    op.process_json(docs[half:])
    rows = op.after_items()
    st = op.stats()
    op.done()
    return rows, st


def test_device_json_extractor_against_the_host_extractor():
    """n1k_push_json through the device extractor (json_extract_kernel: one lane per document, bytes staged in LDS, strings as
    ids of a per-batch table) gives the groups the host's scalar extractor gives — FirstFind (value/parsed.go:159-207), NewValue
    typing (value/value.go:367-430) — over documents that exercise every hand-over to the host, and most documents never
    visit the host."""
    rng = np.random.default_rng(77)
    docs = _tricky_docs(20_000, rng)
    # (every typed value is looked at: s and n as group keys — strings, numbers, arrays by their canonical text — x.y through
    #  SUM / COUNT / COUNTN; MIN / MAX would have to order arrays, which no path of the device does)
    keys = ["(`d`.`s`)", "(`d`.`n`)"]
    aggs = sorted(["count(*)", "sum(((`d`.`x`).`y`))", "count(((`d`.`x`).`y`))", "countn(((`d`.`x`).`y`))"])
    host, hst = _groups_from_json(docs, 0, aggs, keys)
    dev, dst = _groups_from_json(docs, 1, aggs, keys)
    assert hst["json_device_docs"] == 0 and dst["json_device_docs"] > 0.2 * len(docs)  # (most of THESE documents are built to need the host)
    assert dev.nkeys == host.nkeys and len(dev.keys) == len(host.keys)
    hm = {pu._canon_key(k): a for k, a in zip(host.keys, host.aggs)}
    dm = {pu._canon_key(k): a for k, a in zip(dev.keys, dev.aggs)}
    assert hm.keys() == dm.keys()
    for k in hm:
        for i, (x, y) in enumerate(zip(dm[k], hm[k])):
            assert pu.values_match(x, y, 1e-12, float_agg=aggs[i].startswith("sum")), (k, aggs[i], x, y)


def test_device_json_extractor_plain_documents_stay_on_the_device():
    """The synthetic data set's documents need no host help at all, and the groups are the oracle's."""
    n = 40_000
    t = n1o.synth_table(n, k_cat=40, total_rows=20_000)
    docs = _docs_of(t, n)
    cond, keys = "(50 < %s)" % D("price"), [D("cat")]
    aggs = sorted(["count(*)", "sum(%s)" % D("price"), "max(%s)" % D("user_id")])
    ora = n1o.run(t, cond, keys, aggs, threads=2)
    op = query_amd.GpuFilterGroup(query_amd.plan.filter_group_plan(cond, keys, aggs))
    op.process_json(docs)
    gpu = op.after_items()
    st = op.stats()
    op.done()
    pu.assert_same_groups(gpu, ora, aggs=aggs)
    assert st["json_device_docs"] >= 0.99 * n  # (json.dumps prints a few prices with 16 - 17 digits: those documents are the host's)


@pytest.mark.parametrize("device", [1, 0])
def test_malformed_document_is_named_by_either_extractor(device):
    docs = [b'{"s": "a", "n": %d}' % i for i in range(6000)]
    docs[4321] = b'{"s": "a", "n": 1,}'
    op = query_amd.GpuFilterGroup(query_amd.plan.filter_group_plan(None, ["(`d`.`s`)"], ["sum((`d`.`n`))"]), json_device=device, json_device_min_docs=1)
    with pytest.raises(query_amd.N1kError) as ei:
        op.process_json(docs)
    op.done()
    assert ei.value.status == _ffi.INVALID and "document 4321" in ei.value.message


@pytest.mark.parametrize("having,check", [
    ("(33000 < sum(%s))" % D("price"), lambda k, a: a["sum"][1] is not None and a["sum"][1] > 33000),
    ("((count(*) < 700) and (%s is not null))" % D("cat"), lambda k, a: a["count"][1] < 700 and k[0][0] > n1o.T_NULL),
    ("((sum(%s) / count(*)) between 40 and 52)" % D("price"), lambda k, a: 40 <= a["sum"][1] / a["count"][1] <= 52),
    ("(%s = \"cat_3\")" % D("cat"), lambda k, a: k[0][1] == b"cat_3"),
    ("(max(%s) = \"n/a\")" % D("price"), lambda k, a: a["max"][1] == b"n/a"),
])
def test_having_over_the_groups(having, check):
    """planner/build_select_sub.go:295: HAVING is a Filter over the final groups; here its condition (group keys,
    aggregates, arithmetic, logic) is evaluated by the same device predicate code as WHERE."""
    t = n1o.synth_table(60_000, k_cat=90)
    aggs = sorted(["sum(%s)" % D("price"), "count(*)", "max(%s)" % D("price")])
    ora = n1o.run(t, None, [D("cat")], aggs, threads=2)
    gpu, _ = pu.run_gpu(t, None, [D("cat")], aggs, having=having, order=[(D("cat"), False)])
    names = [a.split("(")[0] for a in aggs]
    keep = [(k, a) for k, a in zip(ora.keys, ora.aggs) if check(k, dict(zip(names, a)))]
    assert 0 < len(keep) <= len(ora.keys)
    assert len(gpu.keys) == len(keep)
    assert {pu._canon_key(k) for k in gpu.keys} == {pu._canon_key(k) for k, _ in keep}


def test_having_outside_the_groups_is_unsupported():
    pj = query_amd.plan.filter_group_plan(None, [D("cat")], ["count(*)"], having="(%s < 5)" % D("price"))
    with pytest.raises(query_amd.N1kError) as ei:
        query_amd.GpuFilterGroup(pj)
    assert ei.value.status == _ffi.UNSUPPORTED and "HAVING" in ei.value.message


def test_code_object_cache_refuses_a_damaged_file(tmp_path):
    """Run-time-built kernels are cached on disk (query_amd/jit_cache, or N1K_JIT_CACHE).  A cached object carries a trailer
    (size + two checksums): a file that was truncated, or altered by as little as one byte, is compiled again instead of being
    loaded onto the GPU — and the answer is the oracle's either way."""
    import subprocess
    import sys
    script = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import parity_util as pu
from oracle import n1o
from query_amd import plan
D = lambda *n: plan.field_path("default", *n)
t = n1o.synth_table(50_000, k_cat=30)
cond, keys, aggs = "(%%s <= 50.5)" %% D("price"), [D("cat")], sorted(["avg(%%s)" %% D("price"), "count(*)"])
ora = n1o.run(t, cond, keys, aggs)
gpu, st = pu.run_gpu(t, cond, keys, aggs, jit=2)
pu.assert_same_groups(gpu, ora, aggs=aggs)
assert st["spec_kernel"] == 2, st
print("ok")
''' % (ROOT_DIR, os.path.join(ROOT_DIR, "tests"))
    env = dict(os.environ, N1K_JIT_CACHE=str(tmp_path))

    def run():
        r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]

    run()
    files = sorted(tmp_path.glob("*.co"))
    assert files, "no code object was cached"
    good = files[0].read_bytes()
    assert good[-32:-24] == b"N1KCOv1\0"  # the trailer: magic, size, two checksums
    # one byte flipped in the middle of the object; then a truncated file
    bad = bytearray(good)
    bad[len(bad) // 2] ^= 0x5A
    files[0].write_bytes(bytes(bad))
    run()
    assert files[0].read_bytes() == good  # compiled again and written back
    files[0].write_bytes(good[:len(good) // 3])
    run()
    assert files[0].read_bytes() == good
