"""GpuFilterGroup — the device operator that stands in for

    Parallel{Sequence[Filter, InitialGroup]} -> IntermediateGroup -> FinalGroup

of the reference (execution/filter.go, group_initial.go, group_intermediate.go,
group_final.go).  Method names follow the reference's consumer life cycle
(execution/base.go:485-545): process_items ≙ processItem over a batch,
after_items ≙ afterItems, reopen, send_stop, done.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _ffi
from ._ffi import COL_DICT32, COL_TAGGED64


class N1kError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__("%s: %s" % (_ffi.STATUS_NAMES[status] if 0 <= status < len(_ffi.STATUS_NAMES) else status, message))
        self.status = status
        self.message = message


def device_count() -> int:
    return int(_ffi.lib().n1k_device_count())


@dataclass
class GroupRows:
    """Groups as FinalGroup would emit them: per group the key values and the final aggregate values
    in plan order; values are (tag, python value) pairs, strings as bytes."""
    nkeys: int
    naggs: int
    keys: List[tuple]
    aggs: List[tuple]
    partials: List[tuple]
    rep_row: Optional[np.ndarray] = None
    selected: Optional[np.ndarray] = None
    proj: Optional[List[tuple]] = None  # plans with an InitialProject: per group the values of the result terms


def _np_col(col) -> dict:
    """Accepts oracle-independent duck-typed columns: .kind and .tags/.payload or .codes (numpy)."""
    return col


class GpuFilterGroup:
    def __init__(self, plan_json: str, **options):
        self._lib = _ffi.lib()
        self._h = C.c_void_p()
        raw = plan_json.encode() if isinstance(plan_json, str) else plan_json
        st = self._lib.n1k_create(raw, len(raw), C.byref(self._h))
        if st != _ffi.OK:
            raise N1kError(st, (self._lib.n1k_create_error() or b"").decode(errors="replace"))
        self._keep = []
        for k, v in options.items():
            self.set_option(k, v)

    # ------------------------------------------------------------------ plumbing
    def _check(self, st: int):
        if st != _ffi.OK:
            raise N1kError(st, (self._lib.n1k_last_error(self._h) or b"").decode(errors="replace"))

    def set_option(self, name: str, value: int):
        self._check(self._lib.n1k_set_option(self._h, name.encode(), int(value)))

    @property
    def column_paths(self) -> List[str]:
        n = self._lib.n1k_num_columns(self._h)
        return [self._lib.n1k_column_path(self._h, i).decode() for i in range(n)]

    @property
    def aggregate_names(self) -> List[str]:
        n = self._lib.n1k_num_aggregates(self._h)
        return [self._lib.n1k_aggregate_name(self._h, i).decode() for i in range(n)]

    @property
    def projection_terms(self) -> List[tuple]:
        """(expression text, explicit alias or '') of the plan's InitialProject result terms."""
        n = self._lib.n1k_num_projection_terms(self._h)
        return [(self._lib.n1k_projection_expr(self._h, i).decode(), self._lib.n1k_projection_alias(self._h, i).decode())
                for i in range(n)]

    @property
    def num_keys(self) -> int:
        return int(self._lib.n1k_num_keys(self._h))

    # ---------------------------------------------------------------- dictionary
    def intern(self, strings: Sequence[bytes]) -> np.ndarray:
        """Intern byte strings; returns their dictionary codes (uint32)."""
        n = len(strings)
        offs = np.zeros(n + 1, dtype=np.uint64)
        if n:
            offs[1:] = np.cumsum([len(s) for s in strings], dtype=np.uint64)
        blob = b"".join(strings) + b"\0"
        codes = np.zeros(max(n, 1), dtype=np.uint32)
        self._check(self._lib.n1k_dict_intern(self._h, n, offs.ctypes.data, blob, codes.ctypes.data))
        return codes[:n]

    def dict_get(self, code: int) -> bytes:
        p = C.c_void_p()
        ln = C.c_size_t()
        self._check(self._lib.n1k_dict_get(self._h, code, C.byref(p), C.byref(ln)))
        return C.string_at(p.value, ln.value) if ln.value else b""

    # ---------------------------------------------------------------------- data
    def _make_batch(self, nrows: int, cols: Sequence[tuple]):
        """cols: per column (kind, tags_ptr, payload_ptr, codes_ptr) raw addresses."""
        arr = (_ffi.Col * max(len(cols), 1))()
        for i, (kind, tags, payload, codes) in enumerate(cols):
            arr[i].kind = kind
            arr[i].tags = tags
            arr[i].payload = payload
            arr[i].codes = codes
        b = _ffi.Batch()
        b.nrows = nrows
        b.ncols = len(cols)
        b.cols = C.cast(arr, C.POINTER(_ffi.Col))
        return b, arr

    def process_items(self, columns: Sequence[object], dictionary: Optional[Sequence[bytes]] = None,
                      remap: bool = True, rows: Optional[int] = None):
        """Push one host batch.  `columns` follow self.column_paths order and expose .kind plus numpy
        .tags/.payload or .codes; string payloads/codes index `dictionary`, which is interned into the
        handle's dictionary first (codes are remapped when the handle already holds other strings).  A plan that
        names no leaf path (SELECT COUNT(*) ...) has no columns: `rows` is then the batch."""
        nrows = int(rows or 0)
        code_map = None
        if dictionary is not None and len(dictionary):
            code_map = self.intern(list(dictionary))
            if remap and np.array_equal(code_map, np.arange(len(dictionary), dtype=np.uint32)):
                code_map = None
        keep = []
        cols = []
        for c in columns:
            if c.kind == COL_DICT32:
                codes = np.ascontiguousarray(c.codes, dtype=np.uint32)
                if code_map is not None:
                    special = codes >= np.uint32(_ffi.CODE_NULL)
                    codes = np.where(special, codes, code_map[np.minimum(codes, len(code_map) - 1)]).astype(np.uint32)
                keep.append(codes)
                cols.append((COL_DICT32, None, None, codes.ctypes.data))
                nrows = len(codes)
            else:
                tags = np.ascontiguousarray(c.tags, dtype=np.uint8)
                pay = np.ascontiguousarray(c.payload, dtype=np.uint64)
                if code_map is not None:
                    is_str = tags >= _ffi.T_STRING
                    if is_str.any():
                        pay = pay.copy()
                        pay[is_str] = code_map[pay[is_str].astype(np.int64)]
                keep += [tags, pay]
                cols.append((COL_TAGGED64, tags.ctypes.data, pay.ctypes.data, None))
                nrows = len(tags)
        b, arr = self._make_batch(nrows, cols)
        self._check(self._lib.n1k_push_batch(self._h, C.byref(b)))

    def process_device_items(self, nrows: int, cols: Sequence[tuple]):
        """Push one device-resident batch: cols = [(kind, tags_ptr, payload_ptr, codes_ptr)] device addresses."""
        b, arr = self._make_batch(nrows, cols)
        self._keep = [arr]
        self._check(self._lib.n1k_push_device_batch(self._h, C.byref(b)))

    def make_device_batch(self, nrows: int, cols: Sequence[tuple]):
        """Build the n1k_batch of a device-resident batch once; reuse it with process_device_batch()."""
        b, arr = self._make_batch(nrows, cols)
        return (b, arr)

    def process_device_batch(self, batch):
        self._check(self._lib.n1k_push_device_batch(self._h, C.byref(batch[0])))

    # ------------------------------------------------------------ raw documents
    @staticmethod
    def _pack_docs(docs: Sequence[bytes]):
        offsets = np.zeros(len(docs) + 1, dtype=np.uint64)
        np.cumsum([len(d) for d in docs], out=offsets[1:])
        return offsets, b"".join(docs)

    def extract_json(self, docs: Sequence[bytes]) -> List[dict]:
        """n1k_extract_json: the plan's leaf columns of raw JSON documents, as {"tags", "payload"} numpy copies in
        column_paths order (string payloads are codes of the handle's dictionary).  Needs no GPU."""
        offsets, blob = self._pack_docs(docs)
        b = _ffi.Batch()
        self._check(self._lib.n1k_extract_json(self._h, len(docs), offsets.ctypes.data_as(C.POINTER(C.c_uint64)), blob, C.byref(b)))
        out = []
        for c in range(b.ncols):
            col = b.cols[c]
            n = int(b.nrows)
            tags = np.ctypeslib.as_array(C.cast(col.tags, C.POINTER(C.c_uint8)), shape=(n,)).copy() if n else np.zeros(0, np.uint8)
            pay = np.ctypeslib.as_array(C.cast(col.payload, C.POINTER(C.c_uint64)), shape=(n,)).copy() if n else np.zeros(0, np.uint64)
            out.append({"tags": tags, "payload": pay})
        return out

    def process_json(self, docs: Sequence[bytes]):
        """n1k_push_json: ≙ processItem over raw documents (the file datastore's <key>.json contents)."""
        offsets, blob = self._pack_docs(docs)
        self._check(self._lib.n1k_push_json(self._h, len(docs), offsets.ctypes.data_as(C.POINTER(C.c_uint64)), blob))

    def dict_get(self, code: int) -> bytes:
        ptr, ln = C.c_void_p(), C.c_size_t()
        self._check(self._lib.n1k_dict_get(self._h, int(code), C.byref(ptr), C.byref(ln)))
        return C.string_at(ptr.value, ln.value) if ln.value else b""

    def sync(self):
        self._check(self._lib.n1k_sync(self._h))

    # -------------------------------------------------------------------- results
    _VALUE_DT = np.dtype([("tag", "u1"), ("pad", "u1", (7,)), ("v", "<u8")])
    _PARTIAL_DT = np.dtype([("count", "<i8"), ("isum", "<i8"), ("fsum", "<f8"), ("int_exact", "u1"), ("has_float", "u1"),
                            ("pad", "u1", (6,)), ("ext_tag", "u1"), ("ext_pad", "u1", (7,)), ("ext_v", "<u8"),
                            ("distinct", "<i8")])

    def _finish(self):
        res = _ffi.Result()
        self._check(self._lib.n1k_finish(self._h, C.byref(res)))
        return res

    def run_device_batch_raw(self, batch) -> dict:
        """n1k_run_device_batch: reopen + process_device_batch + after_items_raw in ONE call through the ABI."""
        res = _ffi.Result()
        self._check(self._lib.n1k_run_device_batch(self._h, C.byref(batch[0]), C.byref(res)))
        return self.after_items_raw(res)

    def after_items_raw(self, res=None) -> dict:
        """FinalGroup output as numpy arrays (copies): keys/aggs are structured (tag, v) arrays of shape
        [ngroups, nkeys] / [ngroups, naggs]; string values are dictionary codes."""
        if res is None:
            res = self._finish()
        ng, nk, na = int(res.ngroups), int(res.nkeys), int(res.naggs)

        def arr(ptr, count, dt):
            if not count or not ptr:
                return np.zeros(0, dtype=dt)
            # one memcpy out of the handle-owned result (valid until the next finish/reset/destroy)
            nbytes = count * dt.itemsize
            if nbytes > (1 << 20):  # big results (Filter-only ordinals): view + a single copy
                raw = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(nbytes,))
                return raw.view(dt).copy()
            return np.frombuffer(bytearray(C.string_at(ptr, nbytes)), dtype=dt)

        out = {"ngroups": ng, "nkeys": nk, "naggs": na,
               "keys": arr(res.keys, ng * nk, self._VALUE_DT).reshape(ng, nk) if nk else np.zeros((ng, 0), self._VALUE_DT),
               "aggs": arr(res.aggs, ng * na, self._VALUE_DT).reshape(ng, na) if na else np.zeros((ng, 0), self._VALUE_DT),
               "partials": arr(res.partials, ng * na, self._PARTIAL_DT).reshape(ng, na) if na else None,
               "rep_row": arr(res.rep_row, ng, np.dtype("<u8")) if ng else None,
               "selected": arr(res.selected, int(res.nselected), np.dtype("<u8"))}
        npj = int(res.nproj)
        out["nproj"] = npj
        out["proj"] = arr(res.proj, ng * npj, self._VALUE_DT).reshape(ng, npj) if npj else None
        return out

    def order_rows(self, keys: np.ndarray, aggs: np.ndarray) -> dict:
        """n1k_order_rows: the plan's Order / Offset / Limit over result rows gathered from several owners (structured
        (tag, v) arrays [n, nkeys] / [n, naggs] as after_items_raw returns them).  Host only."""
        n = int(keys.shape[0]) if keys.ndim == 2 and keys.shape[1] else int(aggs.shape[0])
        k = np.ascontiguousarray(keys)
        a = np.ascontiguousarray(aggs)
        res = _ffi.Result()
        self._check(self._lib.n1k_order_rows(self._h, n, k.ctypes.data if k.size else None, a.ctypes.data if a.size else None,
                                             C.byref(res)))
        ng, nk, na = int(res.ngroups), int(res.nkeys), int(res.naggs)

        def arr(ptr, count):
            if not count or not ptr:
                return np.zeros(0, dtype=self._VALUE_DT)
            return np.frombuffer(bytearray(C.string_at(ptr, count * self._VALUE_DT.itemsize)), dtype=self._VALUE_DT)

        return {"ngroups": ng, "nkeys": nk, "naggs": na,
                "keys": arr(res.keys, ng * nk).reshape(ng, nk) if nk else np.zeros((ng, 0), self._VALUE_DT),
                "aggs": arr(res.aggs, ng * na).reshape(ng, na) if na else np.zeros((ng, 0), self._VALUE_DT)}

    def _py_values(self, a: np.ndarray, cache: dict) -> list:
        """structured (tag, v) array [n, m] -> list of n tuples of (tag, python value)."""
        tags = a["tag"]
        v = a["v"]
        ints = v.view(np.int64)
        flts = v.view(np.float64)
        rows = []
        for r in range(a.shape[0]):
            row = []
            for c in range(a.shape[1]):
                t = int(tags[r, c])
                if t == _ffi.T_INT:
                    row.append((t, int(ints[r, c])))
                elif t == _ffi.T_FLOAT:
                    row.append((t, float(flts[r, c])))
                elif t >= _ffi.T_STRING:
                    code = int(v[r, c])
                    s = cache.get(code)
                    if s is None:
                        s = cache[code] = self.dict_get(code)
                    row.append((t, s))
                else:
                    row.append((t, None))
            rows.append(tuple(row))
        return rows

    def after_items(self) -> GroupRows:
        raw = self.after_items_raw()
        ng, nk, na = raw["ngroups"], raw["nkeys"], raw["naggs"]
        cache: dict = {}
        out = GroupRows(nk, na, self._py_values(raw["keys"], cache) if ng else [],
                        self._py_values(raw["aggs"], cache) if ng else [], [])
        out.selected = raw["selected"]
        if raw.get("nproj"):
            out.proj = self._py_values(raw["proj"], cache) if ng else []
        if ng and na:
            p = raw["partials"]
            for g in range(ng):
                out.partials.append(tuple(
                    {"count": int(p[g, a]["count"]), "isum": int(p[g, a]["isum"]), "fsum": float(p[g, a]["fsum"]),
                     "int_exact": int(p[g, a]["int_exact"]), "has_float": int(p[g, a]["has_float"]),
                     "distinct": int(p[g, a]["distinct"])} for a in range(na)))
        out.rep_row = raw["rep_row"]
        return out

    def stats(self) -> dict:
        s = _ffi.Stats()
        self._check(self._lib.n1k_get_stats(self._h, C.byref(s)))
        return {f[0]: getattr(s, f[0]) for f in _ffi.Stats._fields_}

    # ----------------------------------------------------------------- life cycle
    def reopen(self):
        self._check(self._lib.n1k_reset(self._h))

    def send_stop(self):
        self._lib.n1k_stop(self._h)

    def done(self):
        if self._h:
            self._lib.n1k_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.done()
        except Exception:
            pass
