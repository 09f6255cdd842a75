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
        super().__init__("%s: %s" % (_ffi.STATUS_NAMES[status] if 0 <= status < 8 else status, message))
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
                      remap: bool = True):
        """Push one host batch.  `columns` follow self.column_paths order and expose .kind plus numpy
        .tags/.payload or .codes; string payloads/codes index `dictionary`, which is interned into the
        handle's dictionary first (codes are remapped when the handle already holds other strings)."""
        nrows = 0
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

    def sync(self):
        self._check(self._lib.n1k_sync(self._h))

    # -------------------------------------------------------------------- results
    def _value(self, v) -> tuple:
        t = v.tag
        if t == _ffi.T_INT:
            return (t, int(v.v.i))
        if t == _ffi.T_FLOAT:
            return (t, float(v.v.f))
        if t in (_ffi.T_STRING, _ffi.T_ARRAY, _ffi.T_OBJECT):
            return (t, self.dict_get(int(v.v.code)))
        return (t, None)

    def after_items(self) -> GroupRows:
        res = _ffi.Result()
        self._check(self._lib.n1k_finish(self._h, C.byref(res)))
        ng, nk, na = int(res.ngroups), int(res.nkeys), int(res.naggs)
        out = GroupRows(nk, na, [], [], [])
        if res.nselected or not ng:
            out.selected = (np.ctypeslib.as_array(res.selected, shape=(int(res.nselected),)).copy()
                            if res.nselected else np.zeros(0, dtype=np.uint64))
        for g in range(ng):
            out.keys.append(tuple(self._value(res.keys[g * nk + k]) for k in range(nk)))
            out.aggs.append(tuple(self._value(res.aggs[g * na + a]) for a in range(na)))
            parts = []
            for a in range(na):
                p = res.partials[g * na + a]
                parts.append({"count": int(p.count), "isum": int(p.isum), "fsum": float(p.fsum),
                              "int_exact": int(p.int_exact), "has_float": int(p.has_float),
                              "extreme": self._value(p.extreme), "distinct": int(p.distinct)})
            out.partials.append(tuple(parts))
        if ng and res.rep_row:
            out.rep_row = np.ctypeslib.as_array(res.rep_row, shape=(ng,)).copy()
        return out

    def stats(self) -> dict:
        s = _ffi.Stats()
        self._check(self._lib.n1k_get_stats(self._h, C.byref(s)))
        return {f[0]: getattr(s, f[0]) for f in _ffi.Stats._fields_ if f[0] != "reserved"}

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
