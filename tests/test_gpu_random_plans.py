"""Differential testing: seeded random plans (conditions, group keys, aggregates from the device subset) over random
tables holding every scalar class, through randomly chosen kernel families, against the oracle."""
import os

import numpy as np
import pytest

import parity_util as pu
import query_amd
from oracle import n1o
from query_amd import _ffi, plan

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _absolute_slack():
    """Random expressions add terms of both signs: results that cancel to ~1e-17 instead of 0 are rounding, in the
    reference as much as here (see parity_util.ABS_TOL)."""
    old, pu.ABS_TOL = pu.ABS_TOL, 1e-9
    yield
    pu.ABS_TOL = old


def D(name):
    return plan.field_path("default", name)


STRS = [b"", b"a", b"ab", b"b", b"n/a", b"zz", b"10", b"\xc3\xa9"]


def make_table(rng, n):
    """a: every class mixed; b: small ints; f: floats (some integral, some huge); s: dictionary coded strings."""
    def mixed(n):
        tags = np.zeros(n, np.uint8)
        pay = np.zeros(n, np.uint64)
        r = rng.integers(0, 100, n)
        ints = r < 35
        tags[ints] = n1o.T_INT
        pay[ints] = rng.integers(-6, 7, int(ints.sum())).astype(np.int64).view(np.uint64)
        fl = (r >= 35) & (r < 60)
        tags[fl] = n1o.T_FLOAT
        pay[fl] = (rng.integers(-8, 9, int(fl.sum())) / 2.0 + 0.25).view(np.uint64)  # never integral: NewValue keeps floats
        st = (r >= 60) & (r < 75)
        tags[st] = n1o.T_STRING
        pay[st] = rng.integers(0, len(STRS), int(st.sum())).astype(np.uint64)
        tags[(r >= 75) & (r < 80)] = n1o.T_TRUE
        tags[(r >= 80) & (r < 85)] = n1o.T_FALSE
        tags[(r >= 85) & (r < 93)] = n1o.T_NULL
        tags[r >= 93] = n1o.T_MISSING
        big = rng.random(n) < 0.01
        tags[big] = n1o.T_INT
        # (one sign, like the huge floats: no cancellation at 2^61 that would make small results order dependent)
        pay[big] = (rng.integers(1, 4, int(big.sum())).astype(np.int64) * np.int64(2 ** 61)).view(np.uint64)
        return tags, pay
    at, ap = mixed(n)
    bt = np.full(n, n1o.T_INT, np.uint8)
    # b >= 0: dividing the 2^61-scale values of `a` by a signed b would give huge terms of both signs (see below)
    bp = rng.integers(0, 12, n).astype(np.int64).view(np.uint64).copy()
    bt[rng.random(n) < 0.03] = n1o.T_NULL
    ft = np.full(n, n1o.T_FLOAT, np.uint8)
    fp = (rng.integers(0, 4000, n) / 8.0 + 0.0625).view(np.uint64).copy()
    huge = rng.random(n) < 0.01
    # huge but of one sign: sums that cancel (1e300 - 1e300 + x) depend on the order of the additions in any engine
    fp[huge] = np.array([1e300, 3e299, 2.5e15], np.float64).view(np.uint64)[rng.integers(0, 3, int(huge.sum()))]
    ft[rng.random(n) < 0.02] = n1o.T_MISSING
    sc = rng.integers(0, len(STRS), n).astype(np.uint32)
    sc[rng.random(n) < 0.04] = 0xFFFFFFFE
    sc[rng.random(n) < 0.03] = 0xFFFFFFFF
    return n1o.Table([n1o.Column(D("a"), n1o.COL_TAGGED64, tags=at, payload=ap),
                      n1o.Column(D("b"), n1o.COL_TAGGED64, tags=bt, payload=bp),
                      n1o.Column(D("f"), n1o.COL_TAGGED64, tags=ft, payload=fp),
                      n1o.Column(D("s"), n1o.COL_DICT32, codes=sc)], STRS)


def rand_operand(rng, depth=0):
    r = rng.integers(0, 10)
    if r < 5 or depth > 1:
        return D(["a", "b", "f", "s"][rng.integers(0, 4)])
    if r < 7:
        return ["3", "-2", "2.5", "0", "\"ab\"", "100"][rng.integers(0, 6)]
    op = ["+", "-", "*", "/"][rng.integers(0, 4)]
    if op == "*":
        # products only of small values: a signed small factor times a huge one gives huge terms of both signs, and
        # sums that cancel at 1e300 / 2^61 are order dependent in any engine (the reference's Parallel copies too)
        small = [D("b"), "3", "2.5", D("b"), "-2"]
        return "(%s * %s)" % (small[rng.integers(0, 5)], small[rng.integers(0, 5)])
    if op == "/":
        # likewise the divisor is never the signed column `a`: 1e300 / -3 and 1e300 / 3 in one sum cancel
        pos = [D("b"), D("f"), "3", "2.5", "100"]
        return "(%s / %s)" % (rand_operand(rng, depth + 1), pos[rng.integers(0, 5)])
    return "(%s %s %s)" % (rand_operand(rng, depth + 1), op, rand_operand(rng, depth + 1))


def rand_cond(rng, depth=0):
    r = rng.integers(0, 12)
    if depth < 2 and r < 3:
        return "(%s %s %s)" % (rand_cond(rng, depth + 1), ["and", "or"][rng.integers(0, 2)], rand_cond(rng, depth + 1))
    if depth < 2 and r == 3:
        return "(not %s)" % rand_cond(rng, depth + 1)
    if r < 8:
        return "(%s %s %s)" % (rand_operand(rng), ["<", "<=", "=", "<", "<="][rng.integers(0, 5)], rand_operand(rng))
    if r == 8:
        return "(%s between %s and %s)" % (rand_operand(rng), ["-1", "0", "1.5"][rng.integers(0, 3)], ["4", "7.25", "\"b\""][rng.integers(0, 3)])
    return "(%s is %s)" % (D(["a", "b", "f", "s"][rng.integers(0, 4)]),
                           ["null", "not null", "missing", "not missing", "valued", "not valued"][rng.integers(0, 6)])


def rand_plan(rng):
    cond = rand_cond(rng) if rng.random() < 0.7 else None
    key_pool = [D("s"), D("b"), D("a"), "(%s %% 3)" % D("b"), D("f")]
    nk = int(rng.integers(0, 3))
    keys = [key_pool[i] for i in sorted(rng.choice(len(key_pool), nk, replace=False))]
    aggs = set()
    for _ in range(int(rng.integers(1, 5))):
        f = ["sum", "avg", "min", "max", "count", "countn", "count"][rng.integers(0, 7)]
        if f == "count" and rng.random() < 0.3:
            aggs.add("count(*)")
            continue
        distinct = "distinct " if f in ("sum", "avg", "count", "countn") and rng.random() < 0.3 else ""
        aggs.add("%s(%s%s)" % (f, distinct, rand_operand(rng)))
    return cond, keys, sorted(aggs)


OPTION_SETS = [{}, {}, {"spec": 0}, {"fast": 0}, {"fast": 0, "agg_mode": 1}, {"agg_mode": 4, "partition_levels": 1},
               {"agg_mode": 4}, {"distinct_levels": 1}, {"distinct_words": 0}, {"block": 512}, {"wide": 0}]


@pytest.mark.parametrize("seed", range(int(os.environ.get("N1K_RANDOM_SEEDS", "300"))))
def test_random_plans_agree_with_the_oracle(seed):
    rng = np.random.default_rng(1000 + seed)
    t = make_table(rng, int(rng.integers(1, 6000)))
    cond, keys, aggs = rand_plan(rng)
    opts = dict(OPTION_SETS[rng.integers(0, len(OPTION_SETS))])
    batches = int(rng.integers(1, 4))
    resident = bool(rng.random() < 0.3)  # columns already in HBM (n1k_push_device_batch) or host buffers (n1k_push_batch)
    try:
        gpu, _ = pu.run_gpu(t, cond, keys, aggs, batches=batches, device_resident=resident, **opts)
    except query_amd.N1kError as e:
        if e.status == _ffi.UNSUPPORTED:
            pytest.skip("outside the device subset: " + e.message)
        if e.status == _ffi.UNSUPPORTED_DATA and "array" in e.message:
            pytest.skip(e.message)
        raise AssertionError("%s | plan: %r %r %r opts %r" % (e, cond, keys, aggs, opts))
    try:
        ora = n1o.run(t, cond, keys, aggs, threads=2)
    except n1o.OracleError as e:
        pytest.skip("outside the oracle's restated subset: %s" % e)
    try:
        pu.assert_same_groups(gpu, ora, aggs=aggs)
    except AssertionError as e:
        raise AssertionError("%s | plan: %r %r %r opts %r batches %d" % (e, cond, keys, aggs, opts, batches))


@pytest.mark.parametrize("seed", range(int(os.environ.get("N1K_RANDOM_SEEDS", "300")) // 3))
def test_random_order_having_limit_agree_with_the_oracle(seed):
    """The grouped tail on top of random plans: HAVING over an aggregate or key, ORDER BY 1-3 terms (keys / aggregates,
    ASC / DESC), OFFSET / LIMIT — against the oracle's groups filtered and sorted here with the reference's collation."""
    rng = np.random.default_rng(77_000 + seed)
    t = make_table(rng, int(rng.integers(50, 4000)))
    cond, keys, aggs = rand_plan(rng)
    aggs = [a for a in aggs if "distinct" not in a or rng.random() < 0.5] or ["count(*)"]
    if not keys:
        keys = [D("b")]
    terms = keys + aggs
    order = [(terms[i], bool(rng.integers(0, 2))) for i in rng.choice(len(terms), int(rng.integers(1, min(3, len(terms)) + 1)), replace=False)]
    limit = int(rng.integers(0, 30)) if rng.random() < 0.7 else None
    offset = int(rng.integers(0, 5)) if rng.random() < 0.4 else None
    having = None
    hv = None
    if rng.random() < 0.5:
        i = int(rng.integers(0, len(terms)))
        c = ["0", "2", "5.5", "\"a\""][rng.integers(0, 4)]
        having = "(%s < %s)" % (c, terms[i])
        hv = (i, c)
    opts = {"topk_min_groups": 1} if (limit and having is None and rng.random() < 0.5) else {}
    try:
        gpu, _ = pu.run_gpu(t, cond, keys, aggs, batches=int(rng.integers(1, 3)), order=order, limit=limit, offset=offset,
                            having=having, **opts)
    except query_amd.N1kError as e:
        if e.status in (_ffi.UNSUPPORTED, _ffi.UNSUPPORTED_DATA):
            pytest.skip(e.message)
        raise AssertionError("%s | %r %r %r order %r having %r" % (e, cond, keys, aggs, order, having))
    try:
        ora = n1o.run(t, cond, keys, aggs, threads=2)
    except n1o.OracleError as e:
        pytest.skip("outside the oracle's restated subset: %s" % e)
    if hv is not None:  # c < term, with Compare's NULL / MISSING propagation: only TRUE keeps the group
        i, c = hv
        cv = {"0": (n1o.T_INT, 0), "2": (n1o.T_INT, 2), "5.5": (n1o.T_FLOAT, 5.5), "\"a\"": (n1o.T_STRING, b"a")}[c]
        kept = [(k, a) for k, a in zip(ora.keys, ora.aggs)
                if (k + a)[i][0] > n1o.T_NULL and pu.collate_values(cv, (k + a)[i]) < 0]
        ora.keys, ora.aggs = [k for k, _ in kept], [a for _, a in kept]
    try:
        pu.assert_ordered_groups(gpu, ora, keys, aggs, order, limit, offset)
    except AssertionError as e:
        raise AssertionError("%s | %r %r %r order %r limit %r offset %r having %r opts %r" % (e, cond, keys, aggs, order, limit, offset, having, opts))


# ---- arithmetic evaluated in registers by the run-time-built scan (fused nodes) ---------------------------------------

def rand_fused_node(rng, budget):
    """An arithmetic expression of at most `budget` nodes over the mixed-type columns; returns (text, nodes used)."""
    leaf = lambda: [D("a"), D("b"), D("f"), D("a"), D("b"), "3", "-2", "2.5", "0", "\"ab\""][rng.integers(0, 10)]
    small = lambda: [D("b"), "3", "2.5", "-2"][rng.integers(0, 4)]
    pos = lambda: [D("b"), D("f"), "3", "2.5"][rng.integers(0, 4)]
    used = 1
    inner = leaf()
    if budget >= 2 and rng.random() < 0.5:
        inner, u = rand_fused_node(rng, budget - 1)
        used += u
    r = rng.integers(0, 9)
    if r == 0: return "(%s + %s)" % (inner, leaf()), used
    if r == 1: return "(%s - %s)" % (leaf(), inner), used
    if r == 2: return "(%s * %s)" % (small(), small()), 1       # (small products only: see rand_operand)
    if r == 3: return "(%s / %s)" % (inner, pos()), used
    if r == 4: return "(%s %% %s)" % (inner, pos()), used
    if r == 5: return "idiv(%s, %s)" % (inner, pos()), used
    if r == 6: return "(-%s)" % inner, used
    if r == 7: return "round(%s, %s)" % (inner, ["1", "0", "-1", D("b")][rng.integers(0, 4)]), used
    return "%s(%s)" % (["abs", "ceil", "floor", "sign", "trunc"][rng.integers(0, 5)], inner), used


def rand_fused_plan(rng):
    budget = 3
    cond = None
    if rng.random() < 0.6:
        e, u = rand_fused_node(rng, 1 if rng.random() < 0.7 else 2)
        budget -= u
        cond = "(%s %s %s)" % (e, ["<", "<=", "="][rng.integers(0, 3)], ["3", "0", "2.5", "-1"][rng.integers(0, 4)])
        if rng.random() < 0.3:
            cond = "(%s and (%s is not null))" % (cond, D("f"))
    keys = [[D("s")], [D("s")], [], [D("b")]][rng.integers(0, 4)]
    if budget >= 2 and rng.random() < 0.15:
        keys = ["(%s %% 3)" % D("b")]   # a computed key: a fused node feeding the open-addressed LDS table
        budget -= 1
    aggs = set()
    while budget > 0 and len(aggs) < 3:
        e, u = rand_fused_node(rng, min(budget, 2))
        budget -= u
        aggs.add("%s(%s)" % (["sum", "avg", "min", "max", "count", "countn"][rng.integers(0, 6)], e))
    aggs.add(["count(*)", "sum(%s)" % D("b"), "max(%s)" % D("a")][rng.integers(0, 3)])
    return cond, keys, sorted(aggs)


@pytest.mark.parametrize("seed", range(int(os.environ.get("N1K_FUSED_SEEDS", "24"))))
def test_random_fused_arithmetic_agrees_with_the_oracle(seed):
    """Bounded plan shapes with <= 3 arithmetic nodes, forced through the run-time-built scan (jit=2): the nodes are
    evaluated in registers (stats.spec_kernel == 3) and must give what expression/arith_*.go gives row by row."""
    rng = np.random.default_rng(77000 + seed)
    t = make_table(rng, int(rng.integers(1, 9000)))
    cond, keys, aggs = rand_fused_plan(rng)
    batches = int(rng.integers(1, 3))
    resident = bool(rng.random() < 0.5)
    try:
        gpu, st = pu.run_gpu(t, cond, keys, aggs, batches=batches, device_resident=resident, jit=2)
    except query_amd.N1kError as e:
        if e.status == _ffi.UNSUPPORTED:
            pytest.skip("outside the device subset: " + e.message)
        raise AssertionError("%s | plan: %r %r %r" % (e, cond, keys, aggs))
    try:
        ora = n1o.run(t, cond, keys, aggs, threads=2)
    except n1o.OracleError as e:
        pytest.skip("outside the oracle's restated subset: %s" % e)
    try:
        pu.assert_same_groups(gpu, ora, aggs=aggs)
    except AssertionError as e:
        raise AssertionError("%s | plan: %r %r %r batches %d kernel %d" % (e, cond, keys, aggs, batches, st["spec_kernel"]))
    # the same plan with derived columns (element-wise arith_kernel per node) must agree too
    gpu2, st2 = pu.run_gpu(t, cond, keys, aggs, batches=batches, device_resident=resident, jit=2, fuse_arith=0)
    pu.assert_same_groups(gpu2, ora, aggs=aggs)
    assert st2["spec_kernel"] != 3
