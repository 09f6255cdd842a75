#!/usr/bin/env python3
"""Build tests/golden/*.json from the reference's own test DATA files.

Run in the build container only (the reference tree does not travel to the GPU
box):  python tests/golden/make_golden.py [/root/reference]

What is copied is data — the documents of the tiny test keyspaces and the
expected `results` of the reference's JSON case files — never source code.
No reference code is imported or executed (it is Go).

For every case the SQL statement is kept as documentation; the plan strings
(`condition`, `group_keys`, `aggregates`) are what planner/build_select_sub.go
:209-296 emits for that statement, written in expression.Stringer syntax
(expression/stringer.go), derived by hand: the keyspace alias is the keyspace
name, WHERE becomes Filter.condition, GROUP BY terms become group_keys and the
aggregates are de-duplicated and sorted by their text
(planner/build_select_sub.go:551-558).  HAVING / projection / ORDER BY / LIMIT
sit downstream of the hot path and are replayed by the test
harness (tests/golden_util.py) from the `post` section.
"""
import glob
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
FS = os.path.join(REF, "test/filestore/json/default")
MS = os.path.join(REF, "test/multistore/test_cases")


def load_keyspace(name):
    docs = []
    for f in sorted(glob.glob(os.path.join(FS, name, "*.json"))):
        with open(f) as fh:
            docs.append({"key": os.path.basename(f)[:-5], "doc": json.load(fh)})
    return docs


def load_inserts(path):
    """INSERT INTO ks (KEY,VALUE) VALUES("k", {json}) statements -> {ks: [docs]}"""
    out = {}
    for st in json.load(open(path)):
        s = st["statements"]
        m = re.match(r'\s*INSERT INTO (\w+)\s*\(KEY,VALUE\)\s*VALUES\(\s*"([^"]+)"\s*,\s*(\{.*\})\s*\)\s*$', s, re.S)
        if not m:
            continue
        out.setdefault(m.group(1), []).append({"key": m.group(2), "doc": json.loads(m.group(3))})
    return out


def keep_fields(docs, fields):
    return [{"key": d["key"], "doc": {k: v for k, v in d["doc"].items() if k in fields}} for d in docs]


def cases_of(path):
    return json.load(open(path))


def F(alias, *path):
    """stringer text of a field path: ((`a`.`b`).`c`)"""
    s = "`%s`" % alias
    for p in path:
        s = "(%s.`%s`)" % (s, p)
    return s


def explain_plans():
    """Every Filter / InitialGroup / grouped-tail subtree of the EXPLAIN results the reference's case files hold, verbatim
    (tests/golden/plans.json): what n1k_create is handed at the boundary.  Per plan tree:
      * each Filter node and each InitialGroup node on its own;
      * each Parallel whose child Sequence holds a Filter and / or an InitialGroup, as it stands;
      * each run [Parallel{..InitialGroup..}, IntermediateGroup, FinalGroup, ...rest of that Sequence], and the same run
        continued by the Order / Offset / Limit / FinalProject siblings that follow the Sequence one level up (the glue
        concatenates them the same way)."""
    files = sorted(glob.glob(os.path.join(REF, "test", "**", "*.json"), recursive=True))
    out, seen = [], set()

    def emit(src, kind, node):
        text = json.dumps(node, sort_keys=True)
        if text in seen:
            return
        seen.add(text)
        out.append({"source": src, "kind": kind, "plan": node})

    def has_group(par):
        ch = par.get("~child", {})
        kids = ch.get("~children", []) if ch.get("#operator") == "Sequence" else [ch]
        return any(k.get("#operator") == "InitialGroup" for k in kids if isinstance(k, dict))

    def has_path_node(par):
        ch = par.get("~child", {})
        kids = ch.get("~children", []) if ch.get("#operator") == "Sequence" else [ch]
        return any(k.get("#operator") in ("InitialGroup", "Filter") for k in kids if isinstance(k, dict))

    def walk(src, node, after):
        """after: the siblings that follow `node` in its parent Sequence"""
        if isinstance(node, list):
            for x in node:
                walk(src, x, [])
            return
        if not isinstance(node, dict):
            return
        op = node.get("#operator")
        if op in ("Filter", "InitialGroup"):
            emit(src, op, node)
        if op == "Parallel" and has_path_node(node):
            emit(src, "Parallel", node)
        if op == "Sequence":
            kids = node.get("~children", [])
            for i, k in enumerate(kids):
                if isinstance(k, dict) and k.get("#operator") == "Parallel" and has_group(k) and i + 2 < len(kids) and \
                        kids[i + 1].get("#operator") == "IntermediateGroup" and kids[i + 2].get("#operator") == "FinalGroup":
                    run = kids[i:]
                    emit(src, "grouped tail", {"#operator": "Sequence", "~children": run})
                    tail = [a for a in after if a.get("#operator") in ("Order", "Offset", "Limit", "FinalProject")]
                    if tail:
                        emit(src, "grouped tail + order", {"#operator": "Sequence", "~children": run + tail})
            for i, k in enumerate(kids):
                walk(src, k, [x for x in kids[i + 1:] if isinstance(x, dict)])
            return
        for k, v in node.items():
            if isinstance(v, (dict, list)):
                walk(src, v, [])

    for f in files:
        try:
            doc = json.load(open(f))
        except Exception:
            continue
        walk(os.path.relpath(f, REF), doc, [])
    with open(os.path.join(OUT, "plans.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote %d EXPLAIN plan subtrees from %d files" % (len(out), len({p["source"] for p in out})))


def main():
    explain_plans()
    data = {}
    for ks in ("catalog", "orders", "user_profile", "jobs", "tags", "contacts", "game"):
        data[ks] = load_keyspace(ks)
    agg = load_inserts(os.path.join(MS, "aggregate_functions/insert.json"))
    data["ms_product"] = keep_fields(agg["product"], {"test_id", "color", "unitPrice", "categories"})
    data["ms_orders"] = agg["orders"]
    ints = load_inserts(os.path.join(MS, "integers/insert.json"))
    data["ms_int_orders"] = ints["orders"]

    gbh = cases_of(os.path.join(FS, "cases/case_group_by_having.json"))
    whr = cases_of(os.path.join(FS, "cases/case_where.json"))
    ms_gbh = cases_of(os.path.join(MS, "aggregate_functions/case_group_by_having.json"))
    ms_dis = cases_of(os.path.join(MS, "aggregate_functions/case_distinct.json"))
    ms_int = cases_of(os.path.join(MS, "integers/case_select.json"))

    cases = []

    def add(src_name, src_cases, idx, keyspace, plan, post):
        c = src_cases[idx]
        cases.append({
            "id": "%s#%d" % (src_name, idx),
            "source": src_name,
            "index": idx,
            "statement": c["statements"],
            "keyspace": keyspace,
            "plan": plan,
            "post": post,
            "results": c["results"],
        })

    # ---------------------------------------------------------------- G1: filestore case_group_by_having.json
    g1 = "filestore/case_group_by_having.json"
    cat_list = F("catalog", "pricing", "list")
    aggs5 = sorted(["min(%s)" % cat_list, "max(%s)" % cat_list, "avg(%s)" % cat_list, "sum(%s)" % cat_list,
                    "count(%s)" % cat_list])
    ix = {a.split("(")[0]: i for i, a in enumerate(aggs5)}
    proj5 = [{"as": n, "agg": ix[n]} for n in ("min", "max", "avg", "sum", "count")]
    add(g1, gbh, 0, "catalog",
        {"condition": None, "group_keys": [F("catalog", "type")], "aggregates": ["count(*)"]},
        {"project": [{"as": "type", "key": 0}, {"as": "count", "agg": 0}], "order": [[{"key": 0}, "asc"]]})
    add(g1, gbh, 1, "catalog", {"condition": None, "group_keys": [], "aggregates": aggs5}, {"project": proj5})
    add(g1, gbh, 2, "catalog", {"condition": None, "group_keys": [F("catalog", "type")], "aggregates": aggs5},
        {"project": [{"as": "type", "key": 0}] + proj5, "order": [[{"key": 0}, "asc"]]})
    add(g1, gbh, 3, "catalog", {"condition": None, "group_keys": [F("catalog", "type")], "aggregates": aggs5},
        {"having": [{"agg": ix["count"]}, ">", 1], "project": [{"as": "type", "key": 0}] + proj5,
         "order": [[{"key": 0}, "asc"]]})
    add(g1, gbh, 5, "orders",
        {"condition": None, "group_keys": ["(%s[1])" % F("orders", "orderlines")], "aggregates": ["count(*)"]},
        {"project": [{"as": "$1", "key": 0}, {"as": "count", "agg": 0}], "order": [[{"agg": 0}, "asc"]]})
    add(g1, gbh, 6, "orders",
        {"condition": None, "group_keys": [F("orders", "orderlines")], "aggregates": ["count(*)"]},
        {"project": [{"as": "orderlines", "key": 0}, {"as": "count", "agg": 0}], "order": [[{"key": 0}, "asc"]]})
    add(g1, gbh, 9, "user_profile",
        {"condition": None,
         "group_keys": [F("user_profile", "personal_details", "state"),
                        F("user_profile", "profile_details", "loyalty", "membership_type")],
         "aggregates": ["count(*)"]},
        {"having": [{"key": 1}, "=", "Gold"],
         "project": [{"as": "state", "key": 0}, {"as": "membership_type", "key": 1}, {"as": "gold_members", "agg": 0}],
         "order": [[{"key": 0}, "asc"]]})
    theme = F("user_profile", "profile_details", "prefs", "ui_theme")
    add(g1, gbh, 10, "user_profile", {"condition": None, "group_keys": [theme], "aggregates": ["count(*)"]},
        {"project": [{"as": "ui_theme", "key": 0}, {"as": "theme_usage", "agg": 0}], "order": [[{"key": 0}, "asc"]]})
    jt = F("jobs", "job_title")
    add(g1, gbh, 11, "jobs",
        {"condition": None, "group_keys": [F("jobs", "join_yr")], "aggregates": ["count(distinct %s)" % jt]},
        {"project": [{"as": "distinct_title_count", "agg": 0}, {"as": "join_yr", "key": 0}],
         "order": [[{"key": 0}, "asc"]]})
    # ARRAY_AGG (algebra/agg_array.go, agg_array_distinct.go): cases 4, 12, 14
    add(g1, gbh, 4, "catalog", {"condition": None, "group_keys": [], "aggregates": ["array_agg(%s)" % F("catalog", "asin")]},
        {"project": [{"as": "agg", "agg": 0}]})
    add(g1, gbh, 12, "jobs",
        {"condition": None, "group_keys": [F("jobs", "join_yr")], "aggregates": ["array_agg(distinct %s)" % jt]},
        {"project": [{"as": "distinct_titles", "agg": 0}, {"as": "join_yr", "key": 0}], "order": [[{"key": 0}, "asc"]]})
    a14 = sorted(["array_agg(distinct %s)" % jt, "array_agg(%s)" % jt])
    add(g1, gbh, 14, "jobs", {"condition": None, "group_keys": [F("jobs", "join_yr")], "aggregates": a14},
        {"project": [{"as": "distinct_titles", "agg": a14.index("array_agg(distinct %s)" % jt)},
                     {"as": "titles", "agg": a14.index("array_agg(%s)" % jt)}, {"as": "join_yr", "key": 0}],
         "order": [[{"key": 0}, "asc"]]})
    a13 = sorted(["count(distinct %s)" % jt, "count(%s)" % jt])
    add(g1, gbh, 13, "jobs", {"condition": None, "group_keys": [F("jobs", "join_yr")], "aggregates": a13},
        {"project": [{"as": "distinct_title_count", "agg": a13.index("count(distinct %s)" % jt)},
                     {"as": "title_count", "agg": a13.index("count(%s)" % jt)}, {"as": "join_yr", "key": 0}],
         "order": [[{"key": 0}, "asc"]]})
    add(g1, gbh, 15, "user_profile", {"condition": None, "group_keys": [theme], "aggregates": ["count(*)"]},
        {"project": [{"as": "ui_theme", "key": 0}, {"as": "theme_usage", "agg": 0}],
         "order": [[{"agg": 0}, "asc"], [{"key": 0}, "asc"]]})

    # ---------------------------------------------------------------- G2: multistore aggregate_functions
    g2 = "multistore/aggregate_functions/case_group_by_having.json"
    o_filter = '(%s = "agg_func")' % F("orders", "test_id")
    p_filter = '(%s = "agg_func")' % F("product", "test_id")
    up = F("product", "unitPrice")
    paggs = sorted(["min(%s)" % up, "max(%s)" % up, "avg(%s)" % up, "sum(%s)" % up, "count(%s)" % up])
    pix = {a.split("(")[0]: i for i, a in enumerate(paggs)}
    pproj = [{"as": "min", "agg": pix["min"]}, {"as": "max", "agg": pix["max"]},
             {"as": "avg", "agg": pix["avg"], "round": 5}, {"as": "sum", "agg": pix["sum"], "round": 5},
             {"as": "count", "agg": pix["count"]}]
    add(g2, ms_gbh, 0, "ms_orders",
        {"condition": o_filter, "group_keys": [F("orders", "custId")], "aggregates": ["count(*)"]},
        {"project": [{"as": "custId", "key": 0}, {"as": "c", "agg": 0}],
         "order": [[{"agg": 0}, "asc"], [{"key": 0}, "asc"]]})
    add(g2, ms_gbh, 1, "ms_product", {"condition": p_filter, "group_keys": [], "aggregates": paggs},
        {"project": pproj})
    add(g2, ms_gbh, 2, "ms_product", {"condition": p_filter, "group_keys": [F("product", "color")], "aggregates": paggs},
        {"project": [{"as": "product_color", "key": 0}] + pproj,
         "order": [[{"agg": pix["min"]}, "asc"], [{"agg": pix["avg"]}, "asc"]], "limit": 5})
    add(g2, ms_gbh, 3, "ms_product", {"condition": p_filter, "group_keys": [F("product", "color")], "aggregates": paggs},
        {"having": [{"agg": pix["count"]}, ">", 34], "project": [{"as": "product_colori", "key": 0}] + pproj,
         "order": [[{"agg": pix["min"]}, "asc"], [{"agg": pix["avg"]}, "asc"]]})
    add(g2, ms_gbh, 5, "ms_orders",
        {"condition": o_filter, "group_keys": ["(%s[1])" % F("orders", "orderlines")], "aggregates": ["count(*)"]},
        {"project": [{"as": "$1", "key": 0}, {"as": "count", "agg": 0}], "order": [[{"agg": 0}, "asc"]]})
    add(g2, ms_gbh, 6, "ms_orders",
        {"condition": o_filter, "group_keys": [F("orders", "orderlines")], "aggregates": ["count(*)"]},
        {"project": [{"as": "orderlines", "key": 0}, {"as": "count", "agg": 0}], "order": [[{"key": 0}, "asc"]]})

    add(g2, ms_gbh, 4, "ms_orders", {"condition": o_filter, "group_keys": [], "aggregates": ["array_agg(%s)" % F("orders", "id")]},
        {"project": [{"as": "$1", "agg": 0}]})

    g2d = "multistore/aggregate_functions/case_distinct.json"
    pc = F("product", "categories")
    add(g2d, ms_dis, 2, "ms_product",
        {"condition": p_filter, "group_keys": [pc], "aggregates": ["array_agg(distinct %s)" % F("product", "color")]},
        {"project": [{"as": "coloroptions", "agg": 0}, {"as": "categories", "key": 0}],
         "order": [[{"key": 0}, "asc"], [{"agg": 0}, "asc"]], "limit": 2})
    add(g2d, ms_dis, 1, "ms_product",
        {"condition": p_filter, "group_keys": [pc], "aggregates": ["count(distinct %s)" % F("product", "color")]},
        {"project": [{"as": "numcolors", "agg": 0}, {"as": "category", "key": 0}],
         "order": [[{"agg": 0}, "asc"], [{"key": 0}, "desc"]], "limit": 3})
    a3 = sorted(["count(distinct %s)" % F("product", "color"), "count(%s)" % pc])
    add(g2d, ms_dis, 3, "ms_product", {"condition": p_filter, "group_keys": [], "aggregates": a3},
        {"project": [{"as": "totcolors", "agg": a3.index("count(distinct %s)" % F("product", "color"))},
                     {"as": "totcategories", "agg": a3.index("count(%s)" % pc)}]})
    add(g2d, ms_dis, 5, "ms_product", {"condition": p_filter, "group_keys": [pc], "aggregates": ["count(*)"]},
        {"having": [{"agg": 0}, "between", [15, 30]],
         "project": [{"as": "CATG", "key": 0}, {"as": "numprods", "agg": 0}],
         "order": [[{"key": 0}, "asc"], [{"agg": 0}, "asc"]], "limit": 3})
    add(g2d, ms_dis, 6, "ms_product", {"condition": p_filter, "group_keys": [pc], "aggregates": ["count(*)"]},
        {"project": [{"as": "CATG", "key": 0}, {"as": "numprods", "agg": 0}],
         "order": [[{"key": 0}, "asc"], [{"agg": 0}, "asc"]], "limit": 3})
    add(g2d, ms_dis, 7, "ms_product", {"condition": p_filter, "group_keys": [pc], "aggregates": ["count(*)"]},
        {"project": [{"as": "CATG", "key": 0}, {"as": "numprods", "agg": 0}],
         "order": [[{"agg": 0}, "asc"], [{"key": 0}, "asc"]], "limit": 3})
    cn = F("orders", "cntn")
    a8 = sorted(["countn(%s)" % cn, "countn(distinct %s)" % cn, "count(%s)" % cn, "count(distinct %s)" % cn])
    add(g2d, ms_dis, 8, "ms_orders",
        {"condition": '(%s = "cntn_agg_func")' % F("orders", "test_id"), "group_keys": [], "aggregates": a8},
        {"project": [{"as": "cntn", "agg": a8.index("countn(%s)" % cn)},
                     {"as": "dcntn", "agg": a8.index("countn(distinct %s)" % cn)},
                     {"as": "cnt", "agg": a8.index("count(%s)" % cn)},
                     {"as": "dcnt", "agg": a8.index("count(distinct %s)" % cn)}]})

    # ---------------------------------------------------------------- G3: multistore integers
    g3 = "multistore/integers/case_select.json"
    ifilter = '((%s = "select_big_int") and (%s = "aggr"))' % (F("orders", "test_id"), F("orders", "type"))
    # 1-3: the 64-bit integers as leaf values of the documents (Filter + projection of document fields; 2 and 3 RAW)
    vfilter = '((%s = "select_big_int") and (%s = "value"))' % (F("orders", "test_id"), F("orders", "type"))
    cases.append({"id": "%s#0" % g3, "source": g3, "index": 0, "statement": ms_int[0]["statements"], "keyspace": "game",
                  "plan": {"exprs": [["$1", "9223372036854775807"], ["$2", "(-9223372036854775807)"]]}, "post": {},
                  "results": ms_int[0]["results"]})
    add(g3, ms_int, 1, "ms_int_orders", {"condition": vfilter, "filter_only": True},
        {"project": [{"as": "big", "doc": ["big"]}, {"as": "little", "doc": ["little"]}]})
    add(g3, ms_int, 2, "ms_int_orders", {"condition": vfilter, "filter_only": True}, {"raw": {"doc": ["big"]}})
    add(g3, ms_int, 3, "ms_int_orders", {"condition": vfilter, "filter_only": True}, {"raw": {"doc": ["little"]}})
    add(g3, ms_int, 4, "ms_int_orders",
        {"condition": ifilter, "group_keys": [F("orders", "type")], "aggregates": ["sum(%s)" % F("orders", "num")]},
        {"project": [{"as": "total", "agg": 0}, {"as": "type", "key": 0}]})
    add(g3, ms_int, 5, "ms_int_orders",
        {"condition": '((%s = "select_big_int") and (90 < %s))' % (F("orders", "test_id"), F("orders", "num")),
         "group_keys": [], "aggregates": ["count(1)"]},
        {"project": [{"as": "total", "agg": 0}]})

    # ---------------------------------------------------------------- G4: filestore case_where.json (Filter only)
    g4 = "filestore/case_where.json"

    def fo(idx, ks, cond, fields, order):
        add(g4, whr, idx, ks, {"condition": cond, "filter_only": True},
            {"project": [{"as": f[-1], "doc": list(f)} for f in fields],
             "order": [[{"doc": list(order)}, "asc"]]})

    bo = F("tags", "banned-on")
    # (case 3: an array-element leaf, expression/nav_element.go:49-65.  Cases 4-7, 10-13, 24-30 use LIKE, LENGTH, ANY /
    #  EVERY ... SATISFIES or array / object constructors: outside the path's expression subset)
    fo(3, "catalog", '((%s[0]) = "Jessica Chastain")' % F("catalog", "details", "actors"), [("details", "actors")], ("details", "actors"))
    cases[-1]["post"]["project"][0]["as"] = "actors"
    fo(0, "tags", "(%s is not missing)" % bo, [("banned-on",)], ("banned-on",))
    fo(1, "tags", "(%s is not null)" % bo, [("banned-on",)], ("banned-on",))
    fo(2, "tags", "(%s is null)" % bo, [("banned-on",)], ("banned-on",))
    fo(8, "contacts", '(%s = "dave")' % F("contact", "name"), [("name",)], ("name",))
    fo(9, "catalog", "(%s = 799)" % F("catalog", "pricing", "list"), [("dimensions", "height")], ("dimensions", "height"))
    so = F("orders", "shipped-on")
    fo(14, "orders", "(%s is not valued)" % so, [("id",)], ("id",))
    fo(15, "orders", "(%s is valued)" % so, [("id",)], ("id",))
    fo(16, "orders", "(%s is not null)" % so, [("id",)], ("id",))
    fo(17, "orders", "(%s is null)" % so, [("id",)], ("id",))
    fo(18, "orders", "(%s is not missing)" % so, [("id",)], ("id",))
    fo(19, "orders", "(%s is missing)" % so, [("id",)], ("id",))
    fo(20, "contacts", '(not (%s = "dave"))' % F("contacts", "name"), [("name",)], ("name",))
    fo(21, "game", "(%s <= 8)" % F("game", "score"), [("score",)], ("score",))
    fo(22, "game", "(10 <= %s)" % F("game", "score"), [("score",)], ("score",))
    fo(23, "contacts", '((%s = "dave") or (%s = "earl"))' % (F("contacts", "name"), F("contacts", "name")),
       [("name",)], ("name",))
    # ---------------------------------------------------------------- G6: filestore case_integer.json (constant expressions)
    # SELECTs without FROM: no Filter / Group operator runs, but the results pin the number semantics the path's
    # expressions share (value/integer.go:266-352: 2^53+1 stays an exact int through + * - and unary minus; IDIV / IMOD
    # by zero are NULL; DIV is float).  Kept as expression cases: text in expression.Stringer syntax -> expected value.
    g6 = "filestore/case_integer.json"
    cint = cases_of(os.path.join(FS, "cases/case_integer.json"))

    def ex(idx, pairs):
        c = cint[idx]
        cases.append({"id": "%s#%d" % (g6, idx), "source": g6, "index": idx, "statement": c["statements"], "keyspace": "game",
                      "plan": {"exprs": [[a, t] for a, t in pairs]}, "post": {}, "results": c["results"]})

    ex(0, [("float64", "9007199254740992"), ("int64", "9007199254740993")])  # the float literal prints as float64 holds it
    ex(1, [("float64", "9007199254740992"), ("int64", "9007199254740993"), ("add", "(9007199254740993 + 0)"),
           ("mult", "(9007199254740993 * 1)"), ("neg", "(-9007199254740993)"), ("sub", "(9007199254740993 - 0)")])
    ex(2, [("idiv", "idiv(5, 2)"), ("div", "(5 / 2)"), ("imod", "imod(5, 2)"), ("idiv_zero", "idiv(5, 0)"),
           ("imod_zero", "imod(5, 0)")])
    # ---------------------------------------------------------------- G7: filestore case_func_num.json (numeric functions)
    # ROUND / TRUNC / ABS / CEIL / FLOOR / SIGN / SQRT (expression/func_num.go) are arithmetic nodes of the path (WHERE and
    # aggregate operands, projection over the groups): the constant cases as expression cases, the per-document ones
    # (`SELECT f(score + 0.5) AS x FROM default:game ORDER BY x`) as "row_expr" cases: one value per document, ascending.
    # (Left out: trigonometry / EXP / LN / POWER / RANDOM / DEGREES — not in the device subset — and the NaN() / PosInf() /
    #  NegInf() argument cases, whose argument functions are not.)
    g7 = "filestore/case_func_num.json"
    cnum = cases_of(os.path.join(FS, "cases/case_func_num.json"))

    def exn(idx, text):
        c = cnum[idx]
        alias = list(c["results"][0].keys())[0]
        cases.append({"id": "%s#%d" % (g7, idx), "source": g7, "index": idx, "statement": c["statements"], "keyspace": "game",
                      "plan": {"exprs": [[alias, text]]}, "post": {}, "results": c["results"]})

    def rown(idx, text):
        c = cnum[idx]
        alias = list(c["results"][0].keys())[0]
        cases.append({"id": "%s#%d" % (g7, idx), "source": g7, "index": idx, "statement": c["statements"], "keyspace": "game",
                      "plan": {"row_expr": [alias, text]}, "post": {}, "results": c["results"]})

    sc = F("game", "score")
    exn(0, "abs(-4.599)")
    exn(1, "abs(0.0)")
    exn(17, "ceil(1.4)")
    rown(18, "ceil((%s + 0.5))" % sc)
    rown(33, "floor((%s + 0.5))" % sc)
    exn(34, "floor(1.7)")
    rown(40, "round((%s + 0.5))" % sc)
    exn(41, "round(1.2343534)")
    exn(42, "round(1.8343534)")
    exn(43, "round(1.8343534, 0)")
    exn(44, "round(1.8343534, 3)")
    exn(45, "round(8.8343534, -1)")
    exn(46, "round(1.8343534, -1)")
    exn(47, "sign(-1034.992445)")
    exn(48, "sign(1034.992445)")
    exn(49, "sign(0.292445)")
    exn(50, "sign(0.0000111)")
    exn(51, "sign(0.0)")
    rown(55, "trunc((%s + 0.5))" % sc)
    rown(56, "sqrt(%s)" % sc)
    exn(57, "sqrt(0)")
    exn(59, "trunc(-2.2544, 2)")
    exn(60, "trunc(0.2544, 3)")
    # ---------------------------------------------------------------- G4b: filestore case_func_comp.json (GREATEST / LEAST)
    # expression/func_comp.go:54-67, 124-138: the largest / smallest argument above NULL by value.Collate — the cross-type
    # collation of the path's comparisons, as an arithmetic node.  All 8 cases: four per document ordered by
    # the value ("row_expr"), two per document ordered by ANOTHER field ("row_expr_by": [alias, text, order path]), two over
    # constants of different types ("Yes" against 99: a string collates above a number).
    g4b = "filestore/case_func_comp.json"
    ccmp = cases_of(os.path.join(FS, "cases/case_func_comp.json"))

    def cmp_case(idx, plan):
        c = ccmp[idx]
        cases.append({"id": "%s#%d" % (g4b, idx), "source": g4b, "index": idx, "statement": c["statements"], "keyspace": "game",
                      "plan": plan, "post": {}, "results": c["results"]})

    gid = F("game", "id")
    cmp_case(0, {"row_expr": ["gr", "greatest(%s, 9)" % sc]})
    cmp_case(1, {"row_expr": ["gr", "greatest(%s, 75)" % sc]})
    cmp_case(2, {"row_expr": ["gr", "least(%s, 11)" % sc]})
    cmp_case(3, {"row_expr": ["le", "least(%s, 5)" % sc]})
    cmp_case(4, {"row_expr_by": ["$1", 'greatest(%s, "indigo")' % gid, gid]})
    cmp_case(5, {"row_expr_by": ["$1", 'least(%s, "indigo")' % gid, gid]})
    cmp_case(6, {"exprs": [["A", 'least("Yes", 99)']]})
    cmp_case(7, {"exprs": [["A", 'greatest("Yes", 99)']]})
    # ---------------------------------------------------------------- what is NOT a fixture, and why (omitted.json)
    # Every case of the source files above is either a fixture or listed here with the construct that puts it outside the
    # path's subset (comparisons, arithmetic, logic, leaf access, the aggregates; no LIKE, no collection
    # predicates, no constructors, no other functions).  tests/test_oracle_golden.py checks that the two lists add up.
    sources = {g1: gbh, g4: whr, g2: ms_gbh, g2d: ms_dis, g3: ms_int, g6: cint, g7: cnum, g4b: ccmp}
    rules = [
        (r"\bLIKE\b", "LIKE (expression/comp_like.go): pattern matching is not a comparison of the path (§8 a5)"),
        (r"\b(ANY|EVERY)\b", "ANY / EVERY ... SATISFIES (expression/coll_any.go, coll_every.go): collection predicates with their own variable scope"),
        (r"\blength\(", "LENGTH() (expression/func_str.go): string functions are outside §8 a6"),
        (r"=\s*[\[{]", "array / object constructor as a comparison operand (expression/cons_array.go, cons_object.go)"),
        (r"SELECT\s+DISTINCT\b", "SELECT DISTINCT is the Distinct operator (execution/distinct.go), not InitialGroup"),
        (r"\b(acos|asin|atan|atan2|cos|sin|tan|PI|E|POWER|LN|EXP|LOG|DEGREES|RADIANS|random)\(", "trigonometry / EXP / LN / LOG / POWER / PI / E / DEGREES / RADIANS / RANDOM (expression/func_num.go): not arithmetic nodes of the path"),
        (r"\b(NaN|PosInf|NegInf)\(", "NaN() / PosInf() / NegInf() argument functions (expression/func_num.go)"),
        (r"sqrt\(-1\)", "the result is NaN, which the reference serialises as the STRING \"NaN\" (value/float.go:31-48): a quirk of marshalling, documented in DESIGN.md"),
        (r"loyalty_score", "a WHERE over nested OR / AND of six terms with object-valued projection: the projection returns whole sub-documents (not a leaf column)"),
    ]
    omitted, totals = [], {}
    included = {(c["source"], c["index"]) for c in cases}
    for src_name, src_cases in sources.items():
        totals[src_name] = len(src_cases)
        for i, c in enumerate(src_cases):
            if (src_name, i) in included:
                continue
            why = [msg for pat, msg in rules if re.search(pat, c["statements"], re.I if pat[0] != "S" else 0)]
            if not why:
                raise SystemExit("no reason on record for leaving out %s#%d: %s" % (src_name, i, c["statements"]))
            omitted.append({"source": src_name, "index": i, "statement": c["statements"], "outside_the_subset_because": why})
    with open(os.path.join(OUT, "omitted.json"), "w") as fh:
        json.dump({"totals": totals, "omitted": omitted}, fh, indent=1, sort_keys=True)
    print("%d cases left out, each with its reason" % len(omitted))
    # contact alias differs in case 8 ("FROM default:contacts AS contact")
    used = sorted({c["keyspace"] for c in cases})
    with open(os.path.join(OUT, "cases.json"), "w") as fh:
        json.dump(cases, fh, indent=1, sort_keys=True)
    for ks in used:
        with open(os.path.join(OUT, "data_%s.json" % ks), "w") as fh:
            json.dump(data[ks], fh, separators=(",", ":"), sort_keys=True)
    print("wrote %d cases, %d keyspaces" % (len(cases), len(used)))


if __name__ == "__main__":
    main()
