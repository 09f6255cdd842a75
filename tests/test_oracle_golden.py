"""Pin the CPU oracle against the reference's own golden case files.

These are the vectors that make parity claims meaningful: the expected `results`
come verbatim from the reference's test/filestore and test/multistore case
files; the documents are the reference's test data.
"""
import pytest

import golden_util as gu
from oracle import n1o

CASES = gu.load_cases()


@pytest.mark.parametrize("case", CASES, ids=[c["id"] for c in CASES])
@pytest.mark.parametrize("threads", [1, 3])
def test_oracle_matches_reference_golden(case, threads):
    docs = gu.load_docs(case["keyspace"])
    plan = case["plan"]
    if "exprs" in plan:  # constant expressions (case_integer.json): one row, one value per term
        one = gu.build_table(docs[:1], [], strings=[s for _a, text in plan["exprs"] for s in gu.string_constants(text)])
        got = [{alias: gu.decode_value(n1o.eval_expr(one, text)[0]) for alias, text in plan["exprs"]}]
        assert gu.same_json(got, case["results"]), (got, case["results"])
        return
    if "row_expr" in plan:  # one value per document, ascending (case_func_num.json: SELECT f(score + 0.5) AS x ... ORDER BY x)
        alias, text = plan["row_expr"]
        table = gu.build_table(docs, gu.leaf_paths({"condition": None, "group_keys": [text], "aggregates": []}))
        got = gu.sorted_values(alias, [gu.decode_value(tv) for tv in n1o.eval_expr(table, text)])
        assert gu.same_json(got, case["results"]), (got, case["results"])
        return
    if "row_expr_by" in plan:  # one value per document, in the order of another field (case_func_comp.json: ... ORDER BY id)
        alias, text, by = plan["row_expr_by"]
        table = gu.build_table(docs, gu.leaf_paths({"condition": None, "group_keys": [text, by], "aggregates": []}), strings=gu.string_constants(text))
        vals = [gu.decode_value(tv) for tv in n1o.eval_expr(table, text)]
        keys = [gu.decode_value(tv) for tv in n1o.eval_expr(table, by)]
        got = gu.values_ordered_by(alias, vals, keys)
        assert gu.same_json(got, case["results"]), (got, case["results"])
        return
    table = gu.build_table(docs, gu.leaf_paths(plan))
    if plan.get("filter_only"):
        res = n1o.run(table, plan["condition"], [], [], has_group=False)
        got = gu.replay_filter_post(case, docs, res.selected)
    else:
        res = n1o.run(table, plan["condition"], plan["group_keys"], plan["aggregates"], threads=threads)
        got = gu.replay_post(case, gu.groups_from_result(res))
    assert gu.same_json(got, case["results"]), (got, case["results"])


def test_every_case_of_the_source_files_is_a_fixture_or_has_a_reason():
    """tests/golden/omitted.json (written by make_golden.py next to cases.json): every case of the reference's case files the
    fixtures are cut from is either a fixture or listed with the construct that puts it outside the path's subset."""
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    om = json.load(open(os.path.join(here, "omitted.json")))
    cases = gu.load_cases()
    for src, total in om["totals"].items():
        inc = {c["index"] for c in cases if c["source"] == src}
        out = {o["index"] for o in om["omitted"] if o["source"] == src}
        assert not (inc & out) and inc | out == set(range(total)), src
    assert all(o["outside_the_subset_because"] and o["statement"] for o in om["omitted"])
