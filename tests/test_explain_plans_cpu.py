"""The reference's own EXPLAIN goldens at the boundary: every Filter / InitialGroup / grouped-tail
subtree that test/filestore/.../case_by_id.json, case_orderby_limit.json and test/gsi/test_cases/** hold
(tests/golden/plans.json, extracted verbatim by tests/golden/make_golden.py) is fed to n1k_create as it stands.  It must
either be accepted — with the plan's leaf paths as the columns to bind, in first-use order — or refused as
N1K_UNSUPPORTED (the caller keeps the reference operators); never N1K_INVALID: these are plans the reference's planner
really emits (plan/filter.go:46-53, plan/group.go:54-70, plan/parallel.go:54-67, plan/order.go:51-79, plan/project.go:73-110)."""
import json
import os

import pytest

import golden_util as gu
import query_amd
from query_amd import _ffi

with open(os.path.join(gu.GOLDEN, "plans.json")) as fh:
    PLANS = json.load(fh)


def _first_path_nodes(node, out):
    """Filter (before any group operator) and InitialGroup nodes of a subtree, in plan order."""
    if isinstance(node, list):
        for x in node:
            _first_path_nodes(x, out)
    elif isinstance(node, dict):
        op = node.get("#operator")
        if op == "Filter" and not any(o.get("#operator") == "InitialGroup" for o in out):
            out.append(node)
        elif op == "InitialGroup":
            out.append(node)
        for k in ("~child", "~children"):
            if k in node:
                _first_path_nodes(node[k], out)


def _expected_paths(plan):
    nodes = []
    _first_path_nodes(plan, nodes)
    texts = []
    for n in nodes:
        if n["#operator"] == "Filter":
            texts.append(n["condition"])
        else:
            texts += n.get("group_keys", []) + n.get("aggregates", [])
    paths = []
    for t in texts:
        for p in gu._scan_paths(t):
            if p not in paths and not (p.startswith("`") and p.count("`") == 2 and "." not in p and False):
                paths.append(p)
    return paths


@pytest.mark.parametrize("i", range(len(PLANS)), ids=["%s:%s:%d" % (os.path.basename(p["source"]), p["kind"].replace(" ", "_"), i)
                                                    for i, p in enumerate(PLANS)])
def test_reference_explain_subtrees_are_accepted_or_cleanly_refused(i):
    entry = PLANS[i]
    text = json.dumps(entry["plan"])
    try:
        op = query_amd.GpuFilterGroup(text)
    except query_amd.N1kError as e:
        assert e.status == _ffi.UNSUPPORTED, (e.status, e.message, text)
        assert e.message  # the refusal says why
        return
    try:
        if "cover (" in text or "meta(" in text:
            # leaves the caller evaluates from an index entry / the meta data: each is named by its own text
            assert op.column_paths and all(json.dumps(p)[1:-1] in text for p in op.column_paths), (op.column_paths, text)
        else:
            assert op.column_paths == _expected_paths(entry["plan"]), (op.column_paths, text)
    finally:
        op.done()


def test_some_of_each_kind_run_on_the_device_path():
    """The fixture is not vacuous: plans of the path's own shapes are accepted, not merely refused."""
    accepted = {}
    for entry in PLANS:
        try:
            query_amd.GpuFilterGroup(json.dumps(entry["plan"])).done()
            accepted[entry["kind"]] = accepted.get(entry["kind"], 0) + 1
        except query_amd.N1kError:
            pass
    assert accepted.get("Filter", 0) >= 5 and accepted.get("Parallel", 0) >= 3 and accepted.get("InitialGroup", 0) >= 1
    assert accepted.get("grouped tail", 0) + accepted.get("grouped tail + order", 0) >= 1, accepted
