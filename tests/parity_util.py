"""Helpers shared by the parity tests: run the same plan through the device path (C ABI) and the oracle."""
from __future__ import annotations

import math
from typing import Optional, Sequence

import numpy as np

import query_amd
from query_amd import _ffi, plan as qplan
from oracle import n1o

REL_TOL = 1e-9  # north_star: float SUM/AVG within 1e-9 relative; everything else bit-exact
# Sums of terms of both signs can cancel to (almost) zero, where no relative bound holds — the reference's own result
# then depends on the arrival order of the rows.  The random-plan tests (small magnitudes by construction) allow this
# absolute slack on float SUM / AVG; every other test keeps it at 0.
ABS_TOL = 0.0


def run_gpu(table: n1o.Table, condition: Optional[str], keys: Sequence[str], aggs: Sequence[str], *,
            filter_only: bool = False, batches: int = 1, device_resident: bool = False, order=None, limit=None,
            offset=None, having=None, project=None, **options):
    """Run through libn1k.so.  The table's columns are matched to the plan's leaf paths by name."""
    pj = qplan.filter_group_plan(condition, keys, aggs, filter_only=filter_only, order=order, limit=limit, offset=offset,
                                 having=having, project=project)
    op = query_amd.GpuFilterGroup(pj, **options)
    try:
        by_name = {c.name: c for c in table.columns}
        cols = [by_name[p] for p in op.column_paths]
        n = table.nrows
        step = max(1, (n + batches - 1) // batches)
        lo = 0
        first = True
        keep = []
        while first or lo < n:
            hi = min(n, lo + step)
            part = [_slice(c, lo, hi) for c in cols]
            if device_resident:
                import torch
                codes = op.intern(list(table.dictionary))
                # a caller that interns its dictionary right after n1k_create keeps its own codes
                assert np.array_equal(codes, np.arange(len(table.dictionary), dtype=np.uint32))
                dev = []
                for c in part:
                    if c.kind == n1o.COL_DICT32:
                        t = torch.from_numpy(np.ascontiguousarray(c.codes).view(np.int32)).cuda()
                        keep.append(t)
                        dev.append((_ffi.COL_DICT32, None, None, t.data_ptr()))
                    else:
                        t = torch.from_numpy(np.ascontiguousarray(c.tags)).cuda()
                        p = torch.from_numpy(np.ascontiguousarray(c.payload).view(np.int64)).cuda()
                        keep += [t, p]
                        dev.append((_ffi.COL_TAGGED64, t.data_ptr(), p.data_ptr(), None))
                torch.cuda.synchronize()
                op.process_device_items(hi - lo, dev)
            else:
                op.process_items(part, table.dictionary, rows=hi - lo)  # (rows: the batch of a plan that names no column)
            lo = hi
            first = False
        rows = op.after_items()
        stats = op.stats()
        return rows, stats
    finally:
        op.done()


def _slice(c, lo, hi):
    if c.kind == n1o.COL_DICT32:
        return n1o.Column(c.name, c.kind, codes=c.codes[lo:hi])
    return n1o.Column(c.name, c.kind, tags=c.tags[lo:hi], payload=c.payload[lo:hi])


def values_match(g, o, rel=REL_TOL, float_agg=False, tie_ok=False, folded=False) -> bool:
    """(tag, value) from the device vs the oracle: tags must agree; ints/strings/bools exact; floats within rel.
    For SUM/AVG (float_agg) a float result within rel of an integral one may print as INT on one side and FLOAT on
    the other (value.NewValue folds 74.0 but not 73.99999999999999): compared numerically."""
    if g[0] != o[0]:
        if tie_ok and {g[0], o[0]} == {n1o.T_INT, n1o.T_FLOAT}:
            # MIN / MAX keep the first of two values that collate equal (algebra/agg_min.go:83-94): an int and the
            # float equal to it (0 and 0.0) tie, and which one arrived first depends on the row order — in the
            # reference's Parallel copies as much as here
            return float(g[1]) == float(o[1])
        if float_agg and {g[0], o[0]} == {n1o.T_INT, n1o.T_FLOAT}:
            a, b = float(g[1]), float(o[1])
            return a == b or abs(a - b) <= rel * max(abs(a), abs(b)) or abs(a - b) <= ABS_TOL
        return False
    if tie_ok and g[0] == n1o.T_INT and g[1] != o[1] and min(abs(g[1]), abs(o[1])) > 2 ** 53:
        # value.Collate compares two ints exactly but an int with a float through float64 (value/integer.go:100-118,
        # float.go:106-121): beyond 2^53 that order is not transitive (858 < 859 yet both tie with the float between
        # them), so MIN / MAX over such a mix depend on the arrival order in the reference (and in the oracle's
        # worker threads).  Any int the float64 image cannot tell apart is a correct answer.
        return float(g[1]) == float(o[1])
    if folded and g[0] == n1o.T_INT and g[1] != o[1]:
        # AVG is float64(sum) / float64(count) folded to INT when integral (algebra/agg_avg.go:136-157): an INT here is
        # a float result and carries the float tolerance (which only large magnitudes can use: ints differ by >= 1)
        return abs(g[1] - o[1]) <= rel * max(abs(g[1]), abs(o[1]))
    if g[0] == n1o.T_FLOAT:
        a, b = g[1], o[1]
        if math.isnan(a) or math.isnan(b):
            return math.isnan(a) and math.isnan(b)
        if a == b:
            return True
        return abs(a - b) <= rel * max(abs(a), abs(b)) or (float_agg and abs(a - b) <= ABS_TOL)
    return g[1] == o[1]


def _canon_key(k):
    """Key values as value.NewValue would hold them: an integral float IS the int (value/value.go NewValue folds
    float64 with no fraction into intValue), so (FLOAT, 5.0) from a raw column and (INT, 5) name the same key
    (both print "5"; from 2^53 up the float's shortest digits differ from the integer's and the keys stay apart)."""
    out = []
    for tag, v in k:
        if tag == n1o.T_FLOAT and (math.isnan(v) or math.isinf(v)):
            # NaN / +-Inf marshal as JSON strings (value/float.go:31-48): the group of that string
            tag, v = n1o.T_STRING, (b"NaN" if math.isnan(v) else (b"+Infinity" if v > 0 else b"-Infinity"))
        elif tag == n1o.T_FLOAT and v == int(v):
            # the key's text is FormatFloat(f, 'f', -1): an integral float IS the int of that text — its own value below 2^53,
            # its shortest digits followed by zeros above (2^60 prints 1152921504606847000) — while the text fits an int64
            import decimal
            as_int = int(decimal.Decimal(repr(v)))
            if -2 ** 63 <= as_int < 2 ** 63:
                tag, v = n1o.T_INT, as_int
        out.append((tag, v))
    return tuple(out)


def assert_same_groups(gpu, ora, rel=REL_TOL, aggs: Optional[Sequence[str]] = None):
    assert gpu.nkeys == ora.nkeys and gpu.naggs == ora.naggs
    gmap = {_canon_key(k): a for k, a in zip(gpu.keys, gpu.aggs)}
    omap = {_canon_key(k): a for k, a in zip(ora.keys, ora.aggs)}
    assert len(gmap) == len(gpu.keys), "device emitted a duplicate group (NewDuplicateFinalGroupError)"
    missing = set(omap) - set(gmap)
    extra = set(gmap) - set(omap)
    assert not missing and not extra, ("group sets differ", list(missing)[:5], list(extra)[:5])
    for k, oa in omap.items():
        ga = gmap[k]
        for i, (g, o) in enumerate(zip(ga, oa)):
            fl = bool(aggs) and aggs[i].split("(")[0] in ("sum", "avg")
            mm = bool(aggs) and aggs[i].split("(")[0] in ("min", "max")
            av = bool(aggs) and aggs[i].split("(")[0] == "avg"
            assert values_match(g, o, rel, fl, mm, av), ("aggregate %d of group %r: device %r oracle %r" % (i, k, g, o))


_CLASS = {n1o.T_MISSING: 0, n1o.T_NULL: 1, n1o.T_FALSE: 2, n1o.T_TRUE: 2, n1o.T_INT: 3, n1o.T_FLOAT: 3, n1o.T_STRING: 4,
          n1o.T_ARRAY: 5, n1o.T_OBJECT: 6}


def collate_values(a, b) -> int:
    """value.Collate over (tag, python value) pairs: type order, then numbers / bytes / booleans (value/*.go)."""
    ca, cb = _CLASS[a[0]], _CLASS[b[0]]
    if ca != cb:
        return -1 if ca < cb else 1
    if ca <= 1:
        return 0
    if ca == 2:
        return (a[0] == n1o.T_TRUE) - (b[0] == n1o.T_TRUE)
    x, y = a[1], b[1]
    return (x > y) - (x < y)


def assert_ordered_groups(gpu, ora, keys, aggs, order, limit=None, offset=None, rel=REL_TOL):
    """The device's ORDER BY / OFFSET / LIMIT output against the oracle's groups sorted here with the same terms
    (execution/order.go:121-169).  Rows may trade places only where their sort values tie (sort.Sort is not stable)
    or, for float SUM/AVG terms, agree within `rel`."""
    import functools
    names = list(keys) + list(aggs)

    def term_value(g, text):
        i = names.index(text)
        return g[0][i] if i < len(keys) else g[1][i - len(keys)]

    def cmp(g1, g2):
        for text, desc in order or []:
            c = collate_values(term_value(g1, text), term_value(g2, text))
            if c:
                return -c if desc else c
        return 0

    exp = sorted(zip(ora.keys, ora.aggs), key=functools.cmp_to_key(cmp))
    lo = offset or 0
    hi = len(exp) if limit is None else min(len(exp), lo + limit)
    got = list(zip(gpu.keys, gpu.aggs))
    assert len(got) == max(0, hi - lo)
    omap = {_canon_key(k): a for k, a in zip(ora.keys, ora.aggs)}
    for (k, a) in got:  # every returned group is a group of the oracle, with the same aggregates
        oa = omap[_canon_key(k)]
        for i, (g, o) in enumerate(zip(a, oa)):
            f = aggs[i].split("(")[0]
            assert values_match(g, o, rel, f in ("sum", "avg"), f in ("min", "max"), f == "avg"), (k, i, g, o)
    if not order:
        return
    for i in range(1, len(got)):  # sorted by its own values
        assert cmp(got[i - 1], got[i]) <= 0 or _near(got[i - 1], got[i], order, term_value, rel), ("not sorted at", i)
    for i, (e, g) in enumerate(zip(exp[lo:hi], got)):
        if _canon_key(e[0]) == _canon_key(g[0]):
            continue
        assert _near(e, g, order, term_value, rel), ("row %d differs beyond ties" % i, e, g)


def _near(g1, g2, order, term_value, rel):
    """May two rows trade places?  Term by term: identical values defer to the next term; values that only agree within
    the float tolerance make the order a matter of rounding (either is right); values further apart do not."""
    for text, _ in order:
        a, b = term_value(g1, text), term_value(g2, text)
        if a == b and a[0] != n1o.T_FLOAT:
            continue  # an exact value: the next term decides
        # float sums: g1 carries the oracle's values and g2 the device's; two groups whose sums agree to the last
        # digits on one side may differ in the last bit on the other, and then this term alone orders them
        return values_match(a, b, rel, True, True, True)
    return True
