"""Test harness for the golden cases (tests/golden/cases.json).

Turns the reference's test documents into leaf-path columns (the job of
Fetch + parsedValue.Field on the host, value/parsed.go:159-207), and replays
what sits downstream of the hot path — HAVING, projection, ORDER BY, LIMIT
 — in plain Python so that the reference's expected `results`
can be compared 1:1.
"""
from __future__ import annotations

import functools
import json
import math
import os
import re
import sys
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import n1o  # noqa: E402  (tests may use the oracle)

GOLDEN = os.path.join(ROOT, "tests", "golden")
MISSING = object()  # sentinel for N1QL MISSING


def load_cases() -> List[dict]:
    with open(os.path.join(GOLDEN, "cases.json")) as fh:
        return json.load(fh)


@functools.lru_cache(maxsize=None)
def load_docs(keyspace: str) -> List[dict]:
    with open(os.path.join(GOLDEN, "data_%s.json" % keyspace)) as fh:
        return json.load(fh)


# ------------------------------------------------------------------ leaf paths

_PATH_TOKEN = re.compile(r"`([^`]*)`|\[(-?\d+)\]")


def path_steps(text: str) -> List[Any]:
    """"((`a`.`b`)[1])" -> ['b', 1]  (the root identifier is the keyspace alias)."""
    toks = _PATH_TOKEN.findall(text)
    steps: List[Any] = []
    for name, idx in toks[1:]:
        steps.append(int(idx) if idx != "" else name)
    return steps


def leaf_paths(plan: dict) -> List[str]:
    """All maximal field/element chains in the plan's expression strings, in first-use order."""
    texts = []
    if plan.get("condition"):
        texts.append(plan["condition"])
    texts += plan.get("group_keys", [])
    texts += plan.get("aggregates", [])
    out: List[str] = []
    for t in texts:
        for p in _scan_paths(t):
            if p not in out:
                out.append(p)
    return out


def _scan_paths(s: str) -> List[str]:
    """Find maximal parenthesised chains rooted at a backtick identifier."""
    res = []
    i, n = 0, len(s)
    while i < n:
        if s[i] == '"':
            j = i + 1
            while j < n and s[j] != '"':
                j += 2 if s[j] == "\\" else 1
            i = j + 1
            continue
        if s[i] == "(" or s[i] == "`":
            j = _match_path(s, i)
            if j > i:
                res.append(s[i:j])
                i = j
                continue
        i += 1
    return res


def _match_path(s: str, i: int) -> int:
    """If a path expression starts at i return its end, else i."""
    n = len(s)
    if s[i] == "`":
        j = s.index("`", i + 1) + 1
        return j
    if s[i] != "(":
        return i
    j = _match_path(s, i + 1) if i + 1 < n and s[i + 1] in "(`" else i + 1
    if j == i + 1:
        return i
    if j < n and s[j] == "." and j + 1 < n and s[j + 1] == "`":
        k = s.index("`", j + 2) + 1
        if k < n and s[k] == ")":
            return k + 1
        return i
    if j < n and s[j] == "[":
        m = re.match(r"\[(-?\d+)\]\)", s[j:])
        if m:
            return j + m.end()
    return i


def get_path(doc: Any, steps: Sequence[Any]) -> Any:
    cur = doc
    for st in steps:
        if isinstance(st, int):
            if not isinstance(cur, list):
                return MISSING
            if st < 0:
                st += len(cur)
            if st < 0 or st >= len(cur):
                return MISSING
            cur = cur[st]
        else:
            if not isinstance(cur, dict) or st not in cur:
                return MISSING
            cur = cur[st]
    return cur


# ------------------------------------------------------------ value encoding

def go_format_float(f: float) -> str:
    """strconv.FormatFloat(f, 'f', -1, 64) (value/float.go:31-48)."""
    if f == 0:
        return "0"
    r = repr(float(f))
    if "e" in r or "E" in r:
        from decimal import Decimal
        return format(Decimal(r), "f")
    if r.endswith(".0"):
        r = r[:-2]
    return r


def fold_number(x):
    """go_json + value.NewValue: int64 literals stay int, integral floats fold to int (value/value.go:375-382)."""
    if isinstance(x, bool):
        return x
    if isinstance(x, int):
        if -2**63 <= x < 2**63:
            return x
        return float(x)
    if isinstance(x, float):
        if math.isfinite(x) and x == math.floor(x) and -2**63 <= x < 2**63:
            return int(x)
        return x
    return x


def canonical_json(v: Any) -> str:
    """value.MarshalJSON of arrays/objects: sorted names, compact (value/object.go:30-78)."""
    if v is None:
        return "null"
    if v is True:
        return "true"
    if v is False:
        return "false"
    if isinstance(v, (int, float)):
        v = fold_number(v)
        return str(v) if isinstance(v, int) else go_format_float(v)
    if isinstance(v, str):
        return json.dumps(v, ensure_ascii=False)
    if isinstance(v, list):
        return "[" + ",".join(canonical_json(x) for x in v) + "]"
    if isinstance(v, dict):
        return "{" + ",".join(json.dumps(k, ensure_ascii=False) + ":" + canonical_json(v[k]) for k in sorted(v)) + "}"
    raise TypeError(type(v))


class Dictionary:
    def __init__(self):
        self.strings: List[bytes] = []
        self.index: Dict[bytes, int] = {}

    def code(self, b: bytes) -> int:
        c = self.index.get(b)
        if c is None:
            c = len(self.strings)
            self.index[b] = c
            self.strings.append(b)
        return c


def encode_value(v: Any, d: Dictionary):
    """python JSON value -> (tag, payload u64)."""
    if v is MISSING:
        return n1o.T_MISSING, 0
    if v is None:
        return n1o.T_NULL, 0
    if v is True:
        return n1o.T_TRUE, 0
    if v is False:
        return n1o.T_FALSE, 0
    if isinstance(v, (int, float)):
        v = fold_number(v)
        if isinstance(v, int):
            return n1o.T_INT, v & 0xFFFFFFFFFFFFFFFF
        return n1o.T_FLOAT, int(np.float64(v).view(np.uint64))
    if isinstance(v, str):
        return n1o.T_STRING, d.code(v.encode())
    if isinstance(v, list):
        return n1o.T_ARRAY, d.code(canonical_json(v).encode())
    if isinstance(v, dict):
        return n1o.T_OBJECT, d.code(canonical_json(v).encode())
    raise TypeError(type(v))


def build_table(docs: Sequence[dict], paths: Sequence[str], strings: Sequence[str] = ()) -> n1o.Table:
    """Documents -> TAGGED64 leaf columns + dictionary (`strings`: string constants of the plan a result may hold)."""
    d = Dictionary()
    for s in strings:
        d.code(s.encode())
    cols = []
    for p in paths:
        steps = path_steps(p)
        tags = np.zeros(len(docs), dtype=np.uint8)
        pay = np.zeros(len(docs), dtype=np.uint64)
        for r, doc in enumerate(docs):
            t, v = encode_value(get_path(doc["doc"], steps), d)
            tags[r] = t
            pay[r] = v
        cols.append(n1o.Column(p, n1o.COL_TAGGED64, tags=tags, payload=pay))
    if not cols:  # count(*)-only plans still need the row count
        cols.append(n1o.Column("`#rows`", n1o.COL_TAGGED64, tags=np.full(len(docs), n1o.T_NULL, np.uint8),
                               payload=np.zeros(len(docs), np.uint64)))
    return n1o.Table(cols, d.strings)


def decode_value(tv):
    """(tag, value) from a result -> python JSON value (MISSING sentinel kept)."""
    t, v = tv
    if t == n1o.T_MISSING:
        return MISSING
    if t == n1o.T_NULL:
        return None
    if t == n1o.T_FALSE:
        return False
    if t == n1o.T_TRUE:
        return True
    if t in (n1o.T_INT, n1o.T_FLOAT):
        return v
    if t == n1o.T_STRING:
        return v.decode()
    return json.loads(v.decode())


# ------------------------------------------------------------------ collation

def type_rank(v) -> int:
    if v is MISSING:
        return 0
    if v is None:
        return 1
    if isinstance(v, bool):
        return 2
    if isinstance(v, (int, float)):
        return 3
    if isinstance(v, str):
        return 4
    if isinstance(v, list):
        return 5
    return 6


def collate(a, b) -> int:
    """Value.Collate for JSON values (value/*.go Collate, array.go:561-574, object.go:511-556)."""
    ra, rb = type_rank(a), type_rank(b)
    if ra != rb:
        return ra - rb
    if ra <= 1:
        return 0
    if ra == 2:
        return (a > b) - (a < b)
    if ra == 3:
        return (a > b) - (a < b)
    if ra == 4:
        ab, bb = a.encode(), b.encode()
        return (ab > bb) - (ab < bb)
    if ra == 5:
        for i, x in enumerate(a):
            if i >= len(b):
                return 1
            c = collate(x, b[i])
            if c:
                return c
        return len(a) - len(b)
    if len(a) != len(b):
        return len(a) - len(b)
    for name in sorted(set(a) | set(b)):
        if name not in a:
            return 1
        if name not in b:
            return -1
        c = collate(a[name], b[name])
        if c:
            return c
    return 0


def round_float(x: float, prec: int) -> float:
    """expression/func_num.go:1715-1736 roundFloat"""
    if math.isnan(x) or math.isinf(x):
        return x
    sign = 1.0
    if x < 0:
        sign, x = -1.0, -x
    pw = math.pow(10, float(prec))
    intermed = x * pw + 0.5
    rounder = math.floor(intermed)
    if rounder == intermed and math.fmod(rounder, 2) != 0:
        rounder -= 1
    return sign * rounder / pw


# ---------------------------------------------------------------- post stages

def _term(spec: dict, keys, aggs, doc=None):
    if "key" in spec:
        return keys[spec["key"]]
    if "agg" in spec:
        v = aggs[spec["agg"]]
        if "round" in spec and isinstance(v, (int, float)) and not isinstance(v, bool):
            v = fold_number(round_float(float(v), spec["round"]))
        return v
    if "doc" in spec:
        return get_path(doc, spec["doc"])
    raise KeyError(spec)


def _having(h, keys, aggs) -> bool:
    lhs = _term(h[0], keys, aggs)
    op, rhs = h[1], h[2]
    if lhs is MISSING or lhs is None:
        return False
    if op == "between":
        return collate(lhs, rhs[0]) >= 0 and collate(lhs, rhs[1]) <= 0
    c = collate(lhs, rhs)
    return {">": c > 0, "<": c < 0, "=": c == 0, ">=": c >= 0, "<=": c <= 0}[op]


def having_text(case: dict) -> Optional[str]:
    """The case's HAVING clause as expression.Stringer text over the plan's keys / aggregates (what the Filter node
    after FinalGroup carries): `x > c` prints as (c < x) (expression/comp_gt.go:15-17)."""
    h = case.get("post", {}).get("having")
    if not h:
        return None
    p = case["plan"]
    lhs = p["aggregates"][h[0]["agg"]] if "agg" in h[0] else p["group_keys"][h[0]["key"]]
    lit = lambda v: json.dumps(v)
    op, rhs = h[1], h[2]
    if op == "between":
        return "(%s between %s and %s)" % (lhs, lit(rhs[0]), lit(rhs[1]))
    if op in (">", ">="):
        return "(%s %s %s)" % (lit(rhs), "<" if op == ">" else "<=", lhs)
    return "(%s %s %s)" % (lhs, op, lit(rhs))


def replay_post(case: dict, groups: Sequence[tuple], having_done: bool = False) -> List[dict]:
    """groups: list of (keys[], aggs[]) python values.  Returns result rows like the reference's."""
    post = case["post"]
    rows = [g for g in groups if having_done or "having" not in post or _having(post["having"], g[0], g[1])]
    if "order" in post:
        def cmp(g1, g2):
            for spec, direction in post["order"]:
                c = collate(_term(spec, g1[0], g1[1]), _term(spec, g2[0], g2[1]))
                if c:
                    return -c if direction == "desc" else c
            return 0
        rows = sorted(rows, key=functools.cmp_to_key(cmp))
    if "limit" in post:
        rows = rows[:post["limit"]]
    out = []
    for keys, aggs in rows:
        r = {}
        for p in post["project"]:
            v = _term(p, keys, aggs)
            if v is not MISSING:  # a MISSING projection term is omitted from the result object
                r[p["as"]] = v
        out.append(r)
    return out


def project_terms(case: dict) -> List[tuple]:
    """The case's SELECT list as InitialProject result terms [(expression.Stringer text, alias)] over the plan's group
    keys and aggregates (plan/project.go:73-110); ROUND(agg, n) prints as round(<agg>, n) (expression/stringer.go)."""
    p = case["plan"]
    out = []
    for t in case["post"]["project"]:
        text = p["group_keys"][t["key"]] if "key" in t else p["aggregates"][t["agg"]]
        if "round" in t:
            text = "round(%s, %d)" % (text, t["round"])
        out.append((text, t["as"]))
    return out


def order_terms(case: dict) -> Optional[List[tuple]]:
    """The case's ORDER BY as Order sort terms [(text, descending)]: a term that the SELECT list aliases is named by its
    alias (`alias`), as the planner leaves it; any other by the key's / aggregate's own text."""
    post, p = case["post"], case["plan"]
    if "order" not in post:
        return None
    out = []
    for spec, direction in post["order"]:
        alias = next((t["as"] for t in post["project"] if all(t.get(k) == spec.get(k) for k in ("key", "agg"))
                      and not t["as"].startswith("$")), None)
        text = "`%s`" % alias if alias else (p["group_keys"][spec["key"]] if "key" in spec else p["aggregates"][spec["agg"]])
        out.append((text, direction == "desc"))
    return out


def rows_from_projection(case: dict, res) -> List[dict]:
    """Result rows of a device run whose plan carried the InitialProject: {alias: value}, MISSING terms left out
    (execution/project_initial.go:118-121 sets a field only for a value; a MISSING field does not marshal)."""
    out = []
    for vals in res.proj:
        r = {}
        for t, tv in zip(case["post"]["project"], vals):
            v = decode_value(tv)
            if v is not MISSING:
                r[t["as"]] = v
        out.append(r)
    return out


def replay_filter_post(case: dict, docs: Sequence[dict], selected: Sequence[int]) -> List[dict]:
    post = case["post"]
    rows = [docs[int(i)]["doc"] for i in selected]
    if "raw" in post:  # SELECT RAW expr: the bare values
        return [v for v in (_term(post["raw"], None, None, d) for d in rows) if v is not MISSING]
    if "order" in post:
        def cmp(d1, d2):
            for spec, direction in post["order"]:
                c = collate(_term(spec, None, None, d1), _term(spec, None, None, d2))
                if c:
                    return -c if direction == "desc" else c
            return 0
        rows = sorted(rows, key=functools.cmp_to_key(cmp))
    out = []
    for d in rows:
        r = {}
        for p in post["project"]:
            v = _term(p, None, None, d)
            if v is not MISSING:
                r[p["as"]] = v
        out.append(r)
    return out


def same_json(a, b, rel=0.0) -> bool:
    """reflect.DeepEqual on decoded JSON, numbers compared by value (the reference compares
    float64-decoded JSON, test/filestore/json_test.go:254-426)."""
    if isinstance(a, bool) or isinstance(b, bool):
        return a is b
    if isinstance(a, (int, float)) and isinstance(b, (int, float)):
        if a == b:
            return True
        return rel > 0 and abs(a - b) <= rel * max(abs(a), abs(b))
    if type(a) != type(b):
        return False
    if isinstance(a, list):
        return len(a) == len(b) and all(same_json(x, y, rel) for x, y in zip(a, b))
    if isinstance(a, dict):
        return set(a) == set(b) and all(same_json(a[k], b[k], rel) for k in a)
    return a == b


def groups_from_result(res) -> List[tuple]:
    return [([decode_value(k) for k in ks], [decode_value(a) for a in ag]) for ks, ag in zip(res.keys, res.aggs)]


def sorted_values(alias: str, values: Sequence[Any]) -> List[dict]:
    """[{alias: v}] in value.Collate order (execution/order.go:121-169): the result rows of SELECT expr AS alias ... ORDER BY alias."""
    import functools
    return [{alias: v} for v in sorted(values, key=functools.cmp_to_key(collate))]


import re as _re


def string_constants(text: str):
    """the "..." constants of an expression text (a result of GREATEST / LEAST may be one of them)"""
    return _re.findall(r'"((?:[^"\\]|\\.)*)"', text)

def values_ordered_by(alias: str, values: Sequence[Any], keys: Sequence[Any]) -> List[dict]:
    """[{alias: v}] in the value.Collate order of `keys`: the result rows of SELECT expr ... ORDER BY another_field."""
    import functools
    order = sorted(range(len(values)), key=functools.cmp_to_key(lambda i, j: collate(keys[i], keys[j])))
    return [{alias: values[i]} for i in order]
