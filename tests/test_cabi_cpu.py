"""CPU-side checks of the product: the C-ABI library loads, exports every declared symbol, parses the
reference's plan JSON and refuses what is outside the device subset.  No compute call needs a GPU here."""
import ctypes as C
import os
import re

import pytest

import query_amd
from query_amd import _ffi, plan

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "n1k.h")).read()
    declared = set(re.findall(r"\b(n1k_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"n1k_status", "n1k_handle"}
    assert declared == set(_ffi.SYMBOLS), (declared ^ set(_ffi.SYMBOLS))
    L = C.CDLL(_ffi.LIB_PATH)
    for s in sorted(declared):
        assert hasattr(L, s), "libn1k.so does not export " + s
    assert _ffi.lib().n1k_abi_version() == 3


def test_plan_json_binding_and_names():
    pj = plan.filter_group_plan("(50 < (`default`.`price`))", ["(`default`.`cat`)"],
                                ["count(*)", "sum((`default`.`price`))"])
    op = query_amd.GpuFilterGroup(pj)
    assert op.column_paths == ["(`default`.`price`)", "(`default`.`cat`)"]
    assert op.aggregate_names == ["count(*)", "sum((`default`.`price`))"]  # keys of the "aggregates" attachment
    assert op.num_keys == 1
    op.done()


def test_golden_explain_plan_shape_is_accepted():
    """The EXPLAIN golden of the reference (case_by_id.json:369-456) nests InitialGroup in Parallel{Sequence}."""
    pj = ('{"#operator":"Parallel","~child":{"#operator":"Sequence","~children":[{"#operator":"InitialGroup",'
          '"aggregates":["count(*)","min((`game`.`score`))"],"group_keys":[]}]}}')
    op = query_amd.GpuFilterGroup(pj)
    assert op.aggregate_names == ["count(*)", "min((`game`.`score`))"] and op.num_keys == 0
    op.done()


@pytest.mark.parametrize("pj,status", [
    ('{"#operator":"Filter","condition":"(length((`a`.`b`)) < 3)"}', _ffi.UNSUPPORTED),
    ('{"#operator":"Fetch","keyspace":"x"}', _ffi.UNSUPPORTED),
    ('{"#operator":"InitialGroup","group_keys":[],"aggregates":["median((`a`.`b`))"]}', _ffi.UNSUPPORTED),
    ('{"#operator":"InitialGroup","group_keys":[],"aggregates":["min(distinct (`a`.`b`))"]}', _ffi.INVALID),
    ('{"#operator":"Filter","condition":"(any x in (`a`.`b`) satisfies x end)"}', _ffi.UNSUPPORTED),
    ('{"#operator":"Filter","condition":"((`a`.`b`) <"}', _ffi.INVALID),
    ('not json', _ffi.INVALID),
])
def test_unsupported_plans_are_refused(pj, status):
    with pytest.raises(query_amd.N1kError) as ei:
        query_amd.GpuFilterGroup(pj)
    assert ei.value.status == status


def test_no_cpu_fallback_without_device():
    if query_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    op = query_amd.GpuFilterGroup(plan.filter_group_plan(None, [], ["count(*)"]))
    with pytest.raises(query_amd.N1kError) as ei:
        op.after_items()
    assert ei.value.status == _ffi.DEVICE_ERROR
    op.done()


def test_dictionary_interning_is_stable():
    op = query_amd.GpuFilterGroup(plan.filter_group_plan('((`d`.`s`) = "x")', [], ["count(*)"]))
    a = op.intern([b"x", b"y", b"", b"x"])
    assert a[0] == a[3] and len({int(a[0]), int(a[1]), int(a[2])}) == 3
    assert op.dict_get(int(a[1])) == b"y" and op.dict_get(int(a[2])) == b""
    op.done()


def test_runtime_specialisation_compiles_without_a_gpu():
    """A plan shape that has no ahead-of-time kernel is instantiated through hiprtc (compile only, gfx950)."""
    import numpy as np
    pj = plan.filter_group_plan("((10 < (`d`.`x`)) and ((`d`.`x`) <= 90.5))", ["(`d`.`k`)"],
                                ["avg((`d`.`x`))", "count(*)", "max((`d`.`y`))"])
    op = query_amd.GpuFilterGroup(pj)
    assert op.column_paths == ["(`d`.`x`)", "(`d`.`k`)", "(`d`.`y`)"]
    kinds = np.array([_ffi.COL_TAGGED64, _ffi.COL_DICT32, _ffi.COL_TAGGED64], dtype=np.uint32)
    log = C.create_string_buffer(4096)
    st = _ffi.lib().n1k_jit_check(op._h, kinds.ctypes.data, 3, log, 4096)
    assert st == _ffi.OK, log.value.decode(errors="replace")
    op.done()


FUSED_ARITH_SHAPES = [
    # (condition, keys, aggregates, column kinds by order of first use)
    ("(100 < ((`d`.`x`) + (`d`.`y`)))", ["(`d`.`k`)"], ["count(*)", "sum(((`d`.`x`) * (`d`.`y`)))"], "TTD"),
    ("(((`d`.`x`) * 2) < 51)", ["(`d`.`k`)"], ["avg(((`d`.`x`) - 10))", "max((-(`d`.`x`)))"], "TD"),
    (None, ["(`d`.`k`)"], ["count(*)", "min(((`d`.`x`) / ((`d`.`y`) - 7)))", "sum(idiv((`d`.`y`), 4))"], "DTT"),
    (None, ["(`d`.`k`)"], ["sum(round(((`d`.`x`) * 1.5), 2))"], "DT"),
]


@pytest.mark.parametrize("case", range(len(FUSED_ARITH_SHAPES)))
def test_fused_arithmetic_shapes_compile_without_a_gpu(case):
    """expression/arith_*.go inside the run-time-built scan: the plan's arithmetic nodes are part of the kernel's shape
    (evaluated in registers, no derived column); the instantiation compiles for gfx950 through hiprtc."""
    import numpy as np
    cond, keys, aggs, kinds = FUSED_ARITH_SHAPES[case]
    op = query_amd.GpuFilterGroup(plan.filter_group_plan(cond, keys, sorted(aggs)))
    assert len(op.column_paths) == len(kinds)
    k = np.array([_ffi.COL_TAGGED64 if c == "T" else _ffi.COL_DICT32 for c in kinds], dtype=np.uint32)
    log = C.create_string_buffer(8192)
    st = _ffi.lib().n1k_jit_check(op._h, k.ctypes.data, len(kinds), log, 8192)
    assert st == _ffi.OK, log.value.decode(errors="replace")
    op.done()


def test_order_limit_nodes_are_part_of_the_plan_contract():
    """plan/order.go:51-79, plan/limit.go:46-53: Order / Offset / Limit after the group operators are accepted when
    their terms are keys or aggregates of the plan; anything else keeps the reference operators (N1K_UNSUPPORTED)."""
    import query_amd
    from query_amd import plan, _ffi
    D = lambda *n: plan.field_path("default", *n)
    keys, aggs = [D("cat"), D("region_id")], ["sum(%s)" % D("price")]
    ok = plan.filter_group_plan("(50 < %s)" % D("price"), keys, aggs, order=[(aggs[0], True), (D("cat"), False)], limit=100, offset=3)
    op = query_amd.GpuFilterGroup(ok)
    assert op.column_paths == [D("price"), D("cat"), D("region_id")]
    op.done()
    query_amd.GpuFilterGroup(plan.filter_group_plan(None, keys, aggs, limit=10)).done()
    bad_plans = [
        plan.filter_group_plan(None, keys, aggs, order=[("avg(%s)" % D("price"), False)]),          # not an aggregate of the plan
        plan.filter_group_plan(None, keys, aggs, order=[(aggs[0], False)], limit=5).replace('"limit": "5"', '"limit": "(1 + 1)"'),
        '{"#operator":"Sequence","~children":[{"#operator":"Order","sort_terms":[{"expr":"count(*)"}]}]}',  # nothing to order
    ]
    assert "(1 + 1)" in bad_plans[1]
    for bad in bad_plans:
        with pytest.raises(query_amd.N1kError) as ei:
            query_amd.GpuFilterGroup(bad)
        assert ei.value.status == _ffi.UNSUPPORTED
    # a FinalGroup that does not repeat the InitialGroup's lists is a malformed plan
    mism = ok.replace('"#operator": "FinalGroup", "aggregates": ["sum', '"#operator": "FinalGroup", "aggregates": ["max')
    with pytest.raises(query_amd.N1kError) as ei:
        query_amd.GpuFilterGroup(mism)
    assert ei.value.status == _ffi.INVALID
