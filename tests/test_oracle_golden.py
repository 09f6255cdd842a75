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
        one = gu.build_table(docs[:1], [])
        got = [{alias: gu.decode_value(n1o.eval_expr(one, text)[0]) for alias, text in plan["exprs"]}]
        assert gu.same_json(got, case["results"]), (got, case["results"])
        return
    if "row_expr" in plan:  # one value per document, ascending (case_func_num.json: SELECT f(score + 0.5) AS x ... ORDER BY x)
        alias, text = plan["row_expr"]
        table = gu.build_table(docs, gu.leaf_paths({"condition": None, "group_keys": [text], "aggregates": []}))
        got = gu.sorted_values(alias, [gu.decode_value(tv) for tv in n1o.eval_expr(table, text)])
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
