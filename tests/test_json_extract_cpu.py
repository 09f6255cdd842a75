"""n1k_extract_json (host C++, no GPU): raw JSON documents -> the plan's leaf columns, against the Python extraction
the golden tests use (golden_util.build_table) on the reference's own data sets, plus the scanner's edge cases."""
import json

import numpy as np
import pytest

import golden_util as gu
import query_amd
from oracle import n1o
from query_amd import _ffi, plan


def D(*names):
    return plan.field_path("default", *names)


def _decode(op, col):
    """(tag, value) per row with dictionary codes resolved to bytes."""
    out = []
    for t, p in zip(col["tags"], col["payload"]):
        t = int(t)
        if t == n1o.T_INT:
            out.append((t, int(np.uint64(p).astype(np.int64))))
        elif t == n1o.T_FLOAT:
            out.append((t, float(np.array([p], np.uint64).view(np.float64)[0])))
        elif t >= n1o.T_STRING:
            out.append((t, op.dict_get(int(p))))
        else:
            out.append((t, None))
    return out


def _expected(table):
    out = []
    for c in table.columns:
        rows = []
        for t, p in zip(c.tags, c.payload):
            t = int(t)
            if t == n1o.T_INT:
                rows.append((t, int(np.uint64(p).astype(np.int64))))
            elif t == n1o.T_FLOAT:
                rows.append((t, float(np.array([p], np.uint64).view(np.float64)[0])))
            elif t >= n1o.T_STRING:
                rows.append((t, bytes(table.dictionary[int(p)])))
            else:
                rows.append((t, None))
        out.append(rows)
    return out


CASES = [c for c in gu.load_cases() if not c["plan"].get("filter_only") and "exprs" not in c["plan"] and "row_expr" not in c["plan"] and "row_expr_by" not in c["plan"]]


@pytest.mark.parametrize("case", CASES, ids=[c["id"] for c in CASES])
def test_extraction_matches_the_golden_tables(case):
    """Every golden case's documents (the reference's test data), serialised as JSON text, give the same columns as
    the Python extraction of the golden tests: types per value.NewValue, canonical texts of arrays / objects."""
    docs = gu.load_docs(case["keyspace"])
    p = case["plan"]
    try:
        op = query_amd.GpuFilterGroup(plan.filter_group_plan(p["condition"], p["group_keys"], p["aggregates"]))
    except query_amd.N1kError as e:
        pytest.skip("plan outside the device subset: " + e.message)
    paths = op.column_paths
    table = gu.build_table(docs, paths)
    raw = [json.dumps(d["doc"], ensure_ascii=False).encode() for d in docs]
    try:
        cols = op.extract_json(raw)
    except query_amd.N1kError as e:
        if e.status == _ffi.UNSUPPORTED:
            pytest.skip(e.message)
        raise
    got = [_decode(op, c) for c in cols]
    exp = _expected(table)[:len(paths)]
    assert got == exp
    op.done()


def _one(doc: bytes, *fields, threads=None):
    paths = [D(*f) if isinstance(f, tuple) else D(f) for f in fields]
    # up to 4 group keys and 8 aggregates per plan: the leaf columns come out in exactly this order
    op = query_amd.GpuFilterGroup(plan.filter_group_plan(None, paths[:4], ["count(%s)" % p for p in paths[4:]] or ["count(*)"]))
    assert op.column_paths == paths
    if threads:
        op.set_option("json_threads", threads)
    cols = op.extract_json([doc])
    r = [_decode(op, c)[0] for c in cols]
    op.done()
    return r


def test_number_typing_follows_newvalue():
    """value/value.go:375-382 + integer.go:354-356: integer literals that fit int64 are INT, floats with no fraction
    fold to INT, everything else FLOAT (G6: 2^53+1 stays exact as an int literal)."""
    doc = b'{"a": 9007199254740993, "b": 1e3, "c": 1.5, "d": -0.0, "e": 9223372036854775808, "f": -9223372036854775808, "g": 12.0, "h": 1E-2}'
    a, b, c, d, e, f, g, h = _one(doc, "a", "b", "c", "d", "e", "f", "g", "h")
    assert a == (n1o.T_INT, 9007199254740993)
    assert b == (n1o.T_INT, 1000)
    assert c == (n1o.T_FLOAT, 1.5)
    assert d == (n1o.T_INT, 0)
    assert e == (n1o.T_FLOAT, 9223372036854775808.0)
    assert f == (n1o.T_INT, -9223372036854775808)
    assert g == (n1o.T_INT, 12)
    assert h == (n1o.T_FLOAT, 0.01)


def test_fields_strings_and_structure():
    doc = ('{"s": "a\\"b\\\\c\\u00e9\\ud83d\\ude00\\n", "t": true, "f": false, "n": null, "dup": 1, "dup": 2, '
           '"o": {"z": [1, 2.50, {"k": "v", "a": null}], "a": 1.0, "e": {}}, "arr": [ ], "nested": {"x": {"y": 7}}, "sc": 5}').encode()
    s, t, f, n, dup, o, arr, y, miss, below_scalar = _one(doc, "s", "t", "f", "n", "dup", "o", "arr", ("nested", "x", "y"), "nope", ("sc", "x"))
    assert s == (n1o.T_STRING, 'a"b\\cé\U0001F600\n'.encode())
    assert t == (n1o.T_TRUE, None) and f == (n1o.T_FALSE, None) and n == (n1o.T_NULL, None)
    assert dup == (n1o.T_INT, 1)  # FirstFind: the first field of a name counts (value/parsed.go:189-193)
    assert o == (n1o.T_OBJECT, b'{"a":1,"e":{},"z":[1,2.5,{"a":null,"k":"v"}]}')  # sorted names, compact, folded numbers
    assert arr == (n1o.T_ARRAY, b"[]")
    assert y == (n1o.T_INT, 7)
    assert miss == (n1o.T_MISSING, None)
    assert below_scalar == (n1o.T_MISSING, None)  # a field of a non-object is MISSING


def test_malformed_documents_are_named():
    op = query_amd.GpuFilterGroup(plan.filter_group_plan(None, [D("a")], ["count(*)"]))
    good = b'{"a": 1}'
    for bad in (b'{"a": 1', b'{"a": tru}', b'{"a": 1} x', b'{"a": "\\q"}', b'{a: 1}', b''):
        with pytest.raises(query_amd.N1kError) as ei:
            op.extract_json([good, good, bad, good])
        assert ei.value.status == _ffi.INVALID and "document 2" in ei.value.message
    assert _decode(op, op.extract_json([b'[1, 2]', b'7', good])[0]) == [(n1o.T_MISSING, None), (n1o.T_MISSING, None), (n1o.T_INT, 1)]
    op.done()


def test_deep_nesting_is_bounded_like_the_reference():
    """Go's encoding/json scanner accepts 10000 nested levels and rejects more; the scanner here skips unwanted values
    without recursion (a hostile '[[[[...' document must not overflow a worker thread's stack), and wanted array /
    object values take the same bound."""
    def nest(depth, inner=b"1"):
        return b"[" * depth + inner + b"]" * depth
    a, = _one(b'{"junk": ' + nest(9_990) + b', "a": 5}', "a")
    assert a == (n1o.T_INT, 5)
    a, = _one(b'{"junk": ' + b'{"k":' * 5_000 + b"null" + b"}" * 5_000 + b', "a": "x"}', "a")
    assert a == (n1o.T_STRING, b"x")
    op = query_amd.GpuFilterGroup(plan.filter_group_plan(None, [D("a")], ["count(*)"]))
    for hostile in (b'{"junk": ' + nest(10_050) + b', "a": 5}', b'{"junk": ' + b"[" * 200_000 + b', "a": 5}'):
        with pytest.raises(query_amd.N1kError) as ei:
            op.extract_json([b'{"a": 1}', hostile])
        assert ei.value.status == _ffi.INVALID and "document 1" in ei.value.message
    op.done()
    # a wanted value nested far deeper than the 64 levels of the first version: canonical text, as the reference would key it
    a, = _one(b'{"a": ' + nest(250, b'{"b": 1.0}') + b"}", "a")
    assert a == (n1o.T_ARRAY, nest(250, b'{"b":1}'))
    # ... up to the bound of the (recursive) re-serialiser: a wanted value nested 10 000 deep is refused as data outside the
    # subset, whatever the size of the calling thread's stack — not a crash, and not "invalid JSON" (it is valid)
    op = query_amd.GpuFilterGroup(plan.filter_group_plan(None, [D("a")], ["count(*)"]))
    for deep in (nest(10_000), b'{"a":' * 9_000 + b"1" + b"}" * 9_000):
        with pytest.raises(query_amd.N1kError) as ei:
            op.extract_json([b'{"a": 1}', b'{"a": ' + deep + b"}"])
        assert ei.value.status == _ffi.UNSUPPORTED_DATA and "document 1" in ei.value.message
    op.done()


def test_threads_agree_and_share_one_dictionary():
    rng = np.random.default_rng(2)
    docs = [json.dumps({"cat": "c%d" % rng.integers(0, 50), "price": float(rng.integers(0, 10000)) / 100,
                        "tags": [int(x) for x in rng.integers(0, 3, 2)]}).encode() for _ in range(20_000)]
    pj = plan.filter_group_plan(None, [D("cat"), D("tags")], ["sum(%s)" % D("price")])
    a, b = query_amd.GpuFilterGroup(pj), query_amd.GpuFilterGroup(pj)
    a.set_option("json_threads", 1)
    b.set_option("json_threads", 7)
    ca, cb = a.extract_json(docs), b.extract_json(docs)
    for x, y in zip(ca, cb):
        assert np.array_equal(x["tags"], y["tags"])
    assert [_decode(a, c) for c in ca] == [_decode(b, c) for c in cb]
    a.done()
    b.done()


def _rand_value(rng, depth=0):
    r = rng.integers(0, 12)
    if r < 3:
        return int(rng.integers(-10 ** int(rng.integers(0, 19)), 10 ** int(rng.integers(0, 19)) + 1))
    if r < 5:
        return float(rng.integers(-10 ** 6, 10 ** 6)) / float(10 ** int(rng.integers(0, 8))) * (10.0 ** int(rng.integers(-3, 4)))
    if r < 7:
        alphabet = ["a", "b", " ", "\"", "\\", "\n", "\t", "é", "ü", "€", "\U0001F600", "/", "\x01", "z"]
        return "".join(alphabet[i] for i in rng.integers(0, len(alphabet), int(rng.integers(0, 8))))
    if r == 7:
        return [True, False, None][rng.integers(0, 3)]
    if depth >= 3 or r == 8:
        return int(rng.integers(0, 5))
    if r < 10:
        return [_rand_value(rng, depth + 1) for _ in range(int(rng.integers(0, 4)))]
    return {"k%d" % rng.integers(0, 5): _rand_value(rng, depth + 1) for _ in range(int(rng.integers(0, 4)))}


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("N1K_JSON_SEEDS", "40"))))
def test_random_documents_match_the_python_extraction(seed):
    """Random nested documents (escapes, non-BMP characters, 19-digit ints, exponents, nested paths) through the C++
    scanner against json.loads + the golden tests' typing / canonical-text rules."""
    rng = np.random.default_rng(500 + seed)
    fields = ["a", "b", "c", "n"]
    docs = []
    for _ in range(200):
        d = {}
        for f in fields:
            if rng.random() < 0.85:
                d[f] = _rand_value(rng)
        if rng.random() < 0.5:
            d["n"] = {"x": _rand_value(rng, 2), "y": {"z": _rand_value(rng, 3)}}
        docs.append({"doc": d})
    paths = [D("a"), D("b"), D("c"), D("n", "x")]
    op = query_amd.GpuFilterGroup(plan.filter_group_plan(None, paths, ["count(%s)" % D("n", "y", "z")]))
    paths = op.column_paths
    raw = [json.dumps(d["doc"], ensure_ascii=bool(rng.integers(0, 2)),
                      separators=((",", ":") if rng.integers(0, 2) else (", ", ": "))).encode() for d in docs]
    got = [_decode(op, c) for c in op.extract_json(raw)]
    exp = _expected(gu.build_table(docs, paths))
    for c, (g, e) in enumerate(zip(got, exp)):
        for r, (x, y) in enumerate(zip(g, e)):
            assert x == y, (paths[c], r, raw[r], x, y)
    op.done()


def test_synth_documents_round_trip_through_the_extractor():
    """n1k_synth_documents (bench / test input: the synthetic rows as raw JSON documents) and n1k_extract_json are inverse to
    each other on the rows' leaf values: MISSING leaves the field out, NULL prints null, floats print their shortest
    round-trip digits (strconv.FormatFloat(f, 'f', -1, 64), value/float.go:31-48), integers their own text."""
    import ctypes as C
    n = 30_000
    t = n1o.synth_table(n, k_cat=50)
    by = {c.name: c for c in t.columns}
    cat, price, user, region = by[D("cat")], by[D("price")], by[D("user_id")], by[D("region_id")]
    blob = np.empty(n * 260, dtype=np.uint8)
    offsets = np.empty(n + 1, dtype=np.uint64)
    used = C.c_size_t(0)
    lib = _ffi.lib()
    codes = np.ascontiguousarray(cat.codes)
    st = lib.n1k_synth_documents(n, 7, codes.ctypes.data, price.tags.ctypes.data, price.payload.ctypes.data, user.payload.ctypes.data,
                                 region.payload.ctypes.data, 33, blob.ctypes.data, blob.size, offsets.ctypes.data, C.byref(used))
    assert st == _ffi.OK and int(offsets[n]) == used.value
    assert lib.n1k_synth_documents(n, 7, codes.ctypes.data, price.tags.ctypes.data, price.payload.ctypes.data, user.payload.ctypes.data,
                                   region.payload.ctypes.data, 33, blob.ctypes.data, 1000, offsets.ctypes.data, C.byref(used)) == _ffi.OOM
    raw = blob.tobytes()
    docs = [raw[int(offsets[i]):int(offsets[i + 1])] for i in range(n)]
    first = json.loads(docs[0])
    assert first["id"] == "d7" and first["pad"] == "x" * 33
    op = query_amd.GpuFilterGroup(plan.filter_group_plan(None, [D("cat")], sorted(["sum(%s)" % D("price"), "max(%s)" % D("user_id"), "min(%s)" % D("region_id")])))
    cols = dict(zip(op.column_paths, op.extract_json(docs)))
    # price: tags and payload bits exactly (the text round-trips every float); the "n/a" strings by their bytes
    pt, pp = cols[D("price")]["tags"], cols[D("price")]["payload"]
    assert np.array_equal(pt, price.tags)
    num = (price.tags == n1o.T_INT) | (price.tags == n1o.T_FLOAT)
    assert np.array_equal(pp[num], price.payload[num])
    assert all(op.dict_get(int(c)) == b"n/a" for c in pp[price.tags == n1o.T_STRING][:50])
    assert np.array_equal(cols[D("user_id")]["payload"], user.payload) and np.array_equal(cols[D("region_id")]["payload"], region.payload)
    ct, cp = cols[D("cat")]["tags"], cols[D("cat")]["payload"]
    assert np.array_equal(ct == n1o.T_MISSING, codes == 0xFFFFFFFF) and np.array_equal(ct == n1o.T_NULL, codes == 0xFFFFFFFE)
    for i in np.flatnonzero(ct == n1o.T_STRING)[:200]:
        assert op.dict_get(int(cp[i])) == b"cat_%d" % codes[i]
    op.done()
