"""ctypes wrapper of the CPU oracle (oracle/libn1o.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() — never by the product package (query_amd/).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libn1o.so")

# value tags (include/n1k.h n1k_tag)
T_MISSING, T_NULL, T_FALSE, T_TRUE, T_INT, T_FLOAT, T_STRING, T_ARRAY, T_OBJECT = range(9)
COL_TAGGED64, COL_DICT32 = 0, 1
CODE_MISSING = 0xFFFFFFFF
CODE_NULL = 0xFFFFFFFE


class _Col(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("reserved", C.c_uint32), ("tags", C.c_void_p),
                ("payload", C.c_void_p), ("codes", C.c_void_p)]


class _ValueU(C.Union):
    _fields_ = [("i", C.c_int64), ("f", C.c_double), ("code", C.c_uint64)]


class _Value(C.Structure):
    _fields_ = [("tag", C.c_uint8), ("pad", C.c_uint8 * 7), ("v", _ValueU)]


class _Table(C.Structure):
    _fields_ = [("nrows", C.c_uint64), ("ncols", C.c_uint32), ("dict_n", C.c_uint32),
                ("names", C.POINTER(C.c_char_p)), ("cols", C.POINTER(_Col)),
                ("dict_offsets", C.c_void_p), ("dict_bytes", C.c_void_p)]


class _Result(C.Structure):
    _fields_ = [("ngroups", C.c_uint64), ("nkeys", C.c_uint32), ("naggs", C.c_uint32),
                ("keys", C.POINTER(_Value)), ("aggs", C.POINTER(_Value)),
                ("nselected", C.c_uint64), ("selected", C.POINTER(C.c_uint64)),
                ("rows_filtered_in", C.c_uint64), ("seconds", C.c_double), ("err", C.c_char * 512),
                ("extra_bytes", C.c_void_p), ("extra_offsets", C.POINTER(C.c_uint64)), ("extra_n", C.c_uint32)]


def build(force: bool = False) -> str:
    """Compile oracle/libn1o.so with gcc (building the checker is not using it)."""
    srcs = [os.path.join(_HERE, f) for f in ("n1o_oracle.c", "n1o_synth.c", "n1o.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "n1k.h"))
    if not force and os.path.exists(_LIB_PATH):
        mt = os.path.getmtime(_LIB_PATH)
        if all(os.path.getmtime(s) <= mt for s in srcs):
            return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libn1o.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.n1o_run.restype = C.c_int
        L.n1o_run.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_char_p), C.c_uint32,
                              C.c_int, C.POINTER(_Table), C.c_int, C.POINTER(_Result)]
        L.n1o_free_result.argtypes = [C.POINTER(_Result)]
        L.n1o_eval.restype = C.c_int
        L.n1o_eval.argtypes = [C.c_char_p, C.POINTER(_Table), C.POINTER(_Value), C.c_char_p, C.c_size_t]
        L.n1o_count_scan.restype = C.c_int64
        L.n1o_count_scan.argtypes = [C.c_char_p]
        L.n1o_synth_columns.restype = None
        L.n1o_synth_columns.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]
        L.n1o_zipf_cdf.argtypes = [C.c_uint32, C.c_void_p]
        _lib = L
    return _lib


@dataclass
class Column:
    """One leaf-path column in the include/n1k.h layout (numpy, host)."""
    name: str
    kind: int
    tags: Optional[np.ndarray] = None     # uint8   (TAGGED64)
    payload: Optional[np.ndarray] = None  # uint64  (TAGGED64)
    codes: Optional[np.ndarray] = None    # uint32  (DICT32)

    @property
    def nrows(self) -> int:
        return int(len(self.codes) if self.kind == COL_DICT32 else len(self.tags))


@dataclass
class Table:
    columns: List[Column]
    dictionary: List[bytes] = field(default_factory=list)

    @property
    def nrows(self) -> int:
        return self.columns[0].nrows if self.columns else 0

    def slice(self, lo: int, hi: int) -> "Table":
        cols = []
        for c in self.columns:
            if c.kind == COL_DICT32:
                cols.append(Column(c.name, c.kind, codes=c.codes[lo:hi]))
            else:
                cols.append(Column(c.name, c.kind, tags=c.tags[lo:hi], payload=c.payload[lo:hi]))
        return Table(cols, self.dictionary)


def _marshal_table(t: Table):
    keep = []
    ncols = len(t.columns)
    cols = (_Col * max(ncols, 1))()
    names = (C.c_char_p * max(ncols, 1))()
    for i, c in enumerate(t.columns):
        names[i] = c.name.encode()
        cols[i].kind = c.kind
        if c.kind == COL_DICT32:
            a = np.ascontiguousarray(c.codes, dtype=np.uint32)
            keep.append(a)
            cols[i].codes = a.ctypes.data
        else:
            a = np.ascontiguousarray(c.tags, dtype=np.uint8)
            b = np.ascontiguousarray(c.payload, dtype=np.uint64)
            keep += [a, b]
            cols[i].tags = a.ctypes.data
            cols[i].payload = b.ctypes.data
    offs = np.zeros(len(t.dictionary) + 1, dtype=np.uint64)
    if t.dictionary:
        offs[1:] = np.cumsum([len(s) for s in t.dictionary], dtype=np.uint64)
    blob = b"".join(t.dictionary) + b"\0"
    buf = C.create_string_buffer(blob, len(blob))
    keep += [offs, buf, cols, names]
    tab = _Table()
    tab.nrows = t.nrows
    tab.ncols = ncols
    tab.dict_n = len(t.dictionary)
    tab.names = C.cast(names, C.POINTER(C.c_char_p))
    tab.cols = C.cast(cols, C.POINTER(_Col))
    tab.dict_offsets = offs.ctypes.data
    tab.dict_bytes = C.cast(buf, C.c_void_p).value
    return tab, keep


def _pyvalue(v: _Value, dictionary: Sequence[bytes]):
    """n1k_value -> (tag, python value)."""
    t = v.tag
    if t == T_INT:
        return (t, int(v.v.i))
    if t == T_FLOAT:
        return (t, float(v.v.f))
    if t in (T_STRING, T_ARRAY, T_OBJECT):
        code = int(v.v.code)
        return (t, dictionary[code] if code < len(dictionary) else None)
    return (t, None)


@dataclass
class GroupResult:
    nkeys: int
    naggs: int
    keys: List[tuple]   # per group: tuple of (tag, value)
    aggs: List[tuple]   # per group: tuple of (tag, value)
    selected: Optional[np.ndarray] = None
    rows_passed: int = 0
    seconds: float = 0.0


class OracleError(RuntimeError):
    pass


def run(table: Table, condition: Optional[str], keys: Sequence[str], aggs: Sequence[str], *,
        has_group: bool = True, threads: int = 1) -> GroupResult:
    """Reference semantics of Parallel{[Filter,] InitialGroup} -> IntermediateGroup -> FinalGroup."""
    L = lib()
    tab, keep = _marshal_table(table)
    karr = (C.c_char_p * max(len(keys), 1))(*[k.encode() for k in keys])
    aarr = (C.c_char_p * max(len(aggs), 1))(*[a.encode() for a in aggs])
    res = _Result()
    rc = L.n1o_run(condition.encode() if condition else None, karr, len(keys), aarr, len(aggs),
                   1 if has_group else 0, C.byref(tab), threads, C.byref(res))
    if rc != 0:
        msg = res.err.decode(errors="replace")
        L.n1o_free_result(C.byref(res))
        raise OracleError(msg)
    try:
        ng, nk, na = int(res.ngroups), int(res.nkeys), int(res.naggs)
        out = GroupResult(nk, na, [], [], rows_passed=int(res.rows_filtered_in), seconds=float(res.seconds))
        dictionary = list(table.dictionary)
        for i in range(int(res.extra_n)):  # values the run built (ARRAY_AGG arrays): codes behind the dictionary's
            lo, hi = int(res.extra_offsets[i]), int(res.extra_offsets[i + 1])
            dictionary.append(C.string_at(res.extra_bytes + lo, hi - lo))
        if not has_group:
            out.selected = np.ctypeslib.as_array(res.selected, shape=(int(res.nselected),)).copy() \
                if res.nselected else np.zeros(0, dtype=np.uint64)
            return out
        for g in range(ng):
            out.keys.append(tuple(_pyvalue(res.keys[g * nk + k], dictionary) for k in range(nk)))
            out.aggs.append(tuple(_pyvalue(res.aggs[g * na + a], dictionary) for a in range(na)))
        return out
    finally:
        L.n1o_free_result(C.byref(res))


def eval_expr(table: Table, expr: str) -> List[tuple]:
    L = lib()
    tab, keep = _marshal_table(table)
    out = (_Value * max(table.nrows, 1))()
    err = C.create_string_buffer(512)
    rc = L.n1o_eval(expr.encode(), C.byref(tab), out, err, 512)
    if rc != 0:
        raise OracleError(err.value.decode(errors="replace"))
    return [_pyvalue(out[i], table.dictionary) for i in range(table.nrows)]


def count_scan(path: str) -> int:
    return int(lib().n1o_count_scan(path.encode()))


def zipf_cdf(k: int) -> np.ndarray:
    cdf = np.zeros(k, dtype=np.float64)
    lib().n1o_zipf_cdf(k, cdf.ctypes.data)
    return cdf


def synth_dictionary(k_cat: int) -> List[bytes]:
    """Dictionary of the synthetic data set: cat_0..cat_{k-1}, then "n/a"."""
    return [b"cat_%d" % i for i in range(k_cat)] + [b"n/a"]


def synth_table(nrows: int, *, k_cat: int = 1000, zipf: bool = False, seed: int = 0x5EED0001,
                first_row: int = 0, total_rows: Optional[int] = None, alias: str = "default") -> Table:
    """Synthetic columns of SURVEY.md §8(d), generated by the C generator."""
    total = nrows if total_rows is None else total_rows
    cat = np.zeros(nrows, dtype=np.uint32)
    pt = np.zeros(nrows, dtype=np.uint8)
    pp = np.zeros(nrows, dtype=np.uint64)
    ut = np.zeros(nrows, dtype=np.uint8)
    up = np.zeros(nrows, dtype=np.uint64)
    rt = np.zeros(nrows, dtype=np.uint8)
    rp = np.zeros(nrows, dtype=np.uint64)
    cdf = zipf_cdf(k_cat) if zipf else None
    lib().n1o_synth_columns(seed, first_row, nrows, total, k_cat, cdf.ctypes.data if zipf else None,
                            cat.ctypes.data, pt.ctypes.data, pp.ctypes.data, ut.ctypes.data, up.ctypes.data,
                            rt.ctypes.data, rp.ctypes.data)
    cols = [
        Column("(`%s`.`cat`)" % alias, COL_DICT32, codes=cat),
        Column("(`%s`.`price`)" % alias, COL_TAGGED64, tags=pt, payload=pp),
        Column("(`%s`.`user_id`)" % alias, COL_TAGGED64, tags=ut, payload=up),
        Column("(`%s`.`region_id`)" % alias, COL_TAGGED64, tags=rt, payload=rp),
    ]
    return Table(cols, synth_dictionary(k_cat))
