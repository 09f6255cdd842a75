"""ctypes binding of include/n1k.h (query_amd/libn1k.so).

The library is hand-written HIP + C++; this module only marshals pointers.
It never falls back to a CPU implementation: if libn1k.so is missing the
import fails loudly, and without a GPU every compute call returns
N1K_DEVICE_ERROR.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("N1K_LIB") or os.path.join(_HERE, "libn1k.so")  # N1K_LIB: ablation builds only

# n1k_status
OK, UNSUPPORTED, EVAL_ERROR, DEVICE_ERROR, OOM, STOPPED, INVALID, UNSUPPORTED_DATA, REGION_FULL = range(9)
STATUS_NAMES = ["N1K_OK", "N1K_UNSUPPORTED", "N1K_EVAL_ERROR", "N1K_DEVICE_ERROR", "N1K_OOM", "N1K_STOPPED",
                "N1K_INVALID", "N1K_UNSUPPORTED_DATA", "N1K_REGION_FULL"]

# n1k_tag
T_MISSING, T_NULL, T_FALSE, T_TRUE, T_INT, T_FLOAT, T_STRING, T_ARRAY, T_OBJECT = range(9)
COL_TAGGED64, COL_DICT32 = 0, 1
CODE_MISSING = 0xFFFFFFFF
CODE_NULL = 0xFFFFFFFE
COMM_ID_BYTES = 128
MODE_AUTO, MODE_LDS_HASH, MODE_LDS_DIRECT, MODE_GLOBAL = range(4)


class Col(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("reserved", C.c_uint32), ("tags", C.c_void_p), ("payload", C.c_void_p),
                ("codes", C.c_void_p)]


class Batch(C.Structure):
    _fields_ = [("nrows", C.c_uint64), ("ncols", C.c_uint32), ("reserved", C.c_uint32), ("cols", C.POINTER(Col))]


class ValueU(C.Union):
    _fields_ = [("i", C.c_int64), ("f", C.c_double), ("code", C.c_uint64)]


class Value(C.Structure):
    _fields_ = [("tag", C.c_uint8), ("pad", C.c_uint8 * 7), ("v", ValueU)]


class Partial(C.Structure):
    _fields_ = [("count", C.c_int64), ("isum", C.c_int64), ("fsum", C.c_double), ("int_exact", C.c_uint8),
                ("has_float", C.c_uint8), ("pad", C.c_uint8 * 6), ("extreme", Value), ("distinct", C.c_int64)]


class Result(C.Structure):
    _fields_ = [("ngroups", C.c_uint64), ("nkeys", C.c_uint32), ("naggs", C.c_uint32), ("keys", C.POINTER(Value)),
                ("aggs", C.POINTER(Value)), ("partials", C.POINTER(Partial)), ("rep_row", C.POINTER(C.c_uint64)),
                ("nselected", C.c_uint64), ("selected", C.POINTER(C.c_uint64)),
                ("nproj", C.c_uint32), ("reserved1", C.c_uint32), ("proj", C.POINTER(Value))]


class Stats(C.Structure):
    _fields_ = [("rows_in", C.c_uint64), ("rows_selected", C.c_uint64), ("groups_out", C.c_uint64),
                ("batches", C.c_uint64), ("device_ms", C.c_double), ("bytes_scanned", C.c_uint64),
                ("agg_mode", C.c_uint32), ("spec_kernel", C.c_uint32),
                ("wide_key_values", C.c_uint64), ("distinct_path", C.c_uint32), ("reserved0", C.c_uint32),
                ("topk_candidates", C.c_uint64), ("json_device_docs", C.c_uint64), ("query_ms", C.c_double)]


class SynthSpec(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("first_row", C.c_uint64), ("nrows", C.c_uint64), ("total_rows", C.c_uint64),
                ("k_cat", C.c_uint32), ("zipf", C.c_uint32), ("cat_cdf", C.c_void_p)]


# every symbol include/n1k.h declares (tests check that the .so exports all of them)
SYMBOLS = [
    "n1k_create", "n1k_destroy", "n1k_reset", "n1k_stop", "n1k_last_error", "n1k_create_error", "n1k_num_columns",
    "n1k_column_path", "n1k_num_keys", "n1k_num_aggregates", "n1k_aggregate_name", "n1k_num_projection_terms",
    "n1k_projection_expr", "n1k_projection_alias", "n1k_dict_intern", "n1k_dict_size",
    "n1k_dict_get", "n1k_set_option", "n1k_push_batch", "n1k_extract_json", "n1k_push_json", "n1k_push_device_batch", "n1k_run_device_batch", "n1k_comm_create_loopback", "n1k_sync", "n1k_finish",
    "n1k_get_stats", "n1k_partition_device_batch", "n1k_export_groups", "n1k_order_rows", "n1k_merge_groups", "n1k_synth_columns",
    "n1k_jit_check", "n1k_partial_words", "n1k_partial_region_bytes", "n1k_export_partials_device", "n1k_export_partials_async",
    "n1k_merge_partials_device",
    "n1k_comm_unique_id", "n1k_comm_create", "n1k_comm_destroy", "n1k_comm_last_error", "n1k_comm_rank", "n1k_comm_world",
    "n1k_comm_max_u64", "n1k_exchange_partials", "n1k_exchange_rows", "n1k_gather_groups", "n1k_gather_groups_status", "n1k_rows_step", "n1k_partials_step", "n1k_failure_is_global",
    "n1k_exchange_rows_v", "n1k_exchange_sent_rows", "n1k_comm_max_u64_v", "n1k_rows_step_v",
    "n1k_synth_documents", "n1k_abi_version", "n1k_device_count",
]

_lib = None


def lib():
    """Load libn1k.so (building is __graft_entry__.build()'s / query_amd.build's job)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so.7; two HIP runtimes in one process cannot both own the GPU.  Importing
    # torch first makes libn1k.so's NEEDED libamdhip64.so.7 bind to the copy torch already loaded (same SONAME).
    # A C/Go consumer of the ABI simply uses the system ROCm runtime.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is plumbing, not a requirement of the library
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "query_amd/libn1k.so is missing: run `python -m query_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the device path.")
    L = C.CDLL(LIB_PATH)
    H = C.c_void_p
    L.n1k_create.restype = C.c_int
    L.n1k_create.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(H)]
    L.n1k_destroy.restype = None
    L.n1k_destroy.argtypes = [H]
    L.n1k_reset.restype = C.c_int
    L.n1k_reset.argtypes = [H]
    L.n1k_stop.restype = None
    L.n1k_stop.argtypes = [H]
    L.n1k_last_error.restype = C.c_char_p
    L.n1k_last_error.argtypes = [H]
    L.n1k_create_error.restype = C.c_char_p
    L.n1k_create_error.argtypes = []
    for f in ("n1k_projection_expr", "n1k_projection_alias"):
        getattr(L, f).restype = C.c_char_p
        getattr(L, f).argtypes = [H, C.c_uint32]
    for f in ("n1k_num_columns", "n1k_num_keys", "n1k_num_aggregates", "n1k_dict_size", "n1k_num_projection_terms"):
        getattr(L, f).restype = C.c_uint32
        getattr(L, f).argtypes = [H]
    L.n1k_column_path.restype = C.c_char_p
    L.n1k_column_path.argtypes = [H, C.c_uint32]
    L.n1k_aggregate_name.restype = C.c_char_p
    L.n1k_aggregate_name.argtypes = [H, C.c_uint32]
    L.n1k_dict_intern.restype = C.c_int
    L.n1k_dict_intern.argtypes = [H, C.c_uint32, C.c_void_p, C.c_char_p, C.c_void_p]
    L.n1k_dict_get.restype = C.c_int
    L.n1k_dict_get.argtypes = [H, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.n1k_set_option.restype = C.c_int
    L.n1k_set_option.argtypes = [H, C.c_char_p, C.c_int64]
    L.n1k_push_batch.restype = C.c_int
    L.n1k_push_batch.argtypes = [H, C.POINTER(Batch)]
    L.n1k_push_device_batch.restype = C.c_int
    L.n1k_push_device_batch.argtypes = [H, C.POINTER(Batch)]
    if hasattr(L, "n1k_run_device_batch"):  # (absent from older builds loaded through N1K_LIB for A/B measurements)
        L.n1k_run_device_batch.restype = C.c_int
        L.n1k_run_device_batch.argtypes = [H, C.POINTER(Batch), C.POINTER(Result)]
    if hasattr(L, "n1k_comm_create_loopback"):
        L.n1k_comm_create_loopback.restype = C.c_int
        L.n1k_comm_create_loopback.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.n1k_extract_json.restype = C.c_int
    L.n1k_extract_json.argtypes = [H, C.c_uint64, C.POINTER(C.c_uint64), C.c_char_p, C.POINTER(Batch)]
    L.n1k_push_json.restype = C.c_int
    L.n1k_push_json.argtypes = [H, C.c_uint64, C.POINTER(C.c_uint64), C.c_char_p]
    L.n1k_sync.restype = C.c_int
    L.n1k_sync.argtypes = [H]
    L.n1k_finish.restype = C.c_int
    L.n1k_finish.argtypes = [H, C.POINTER(Result)]
    L.n1k_get_stats.restype = C.c_int
    L.n1k_get_stats.argtypes = [H, C.POINTER(Stats)]
    L.n1k_partition_device_batch.restype = C.c_int
    L.n1k_partition_device_batch.argtypes = [H, C.POINTER(Batch), C.c_uint32, C.c_uint64, C.POINTER(Col), C.c_void_p]
    L.n1k_export_groups.restype = C.c_int
    L.n1k_export_groups.argtypes = [H, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.n1k_merge_groups.restype = C.c_int
    L.n1k_merge_groups.argtypes = [H, C.c_void_p, C.c_size_t]
    L.n1k_order_rows.restype = C.c_int
    L.n1k_order_rows.argtypes = [H, C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(Result)]
    L.n1k_jit_check.restype = C.c_int
    L.n1k_jit_check.argtypes = [H, C.c_void_p, C.c_uint32, C.c_char_p, C.c_size_t]
    L.n1k_partial_words.restype = C.c_uint32
    L.n1k_partial_words.argtypes = [H]
    L.n1k_partial_region_bytes.restype = C.c_uint64
    L.n1k_partial_region_bytes.argtypes = [H, C.c_uint64]
    L.n1k_export_partials_device.restype = C.c_int
    L.n1k_export_partials_device.argtypes = [H, C.c_uint32, C.c_uint64, C.c_void_p]
    L.n1k_export_partials_async.restype = C.c_int
    L.n1k_export_partials_async.argtypes = [H, C.c_uint32, C.c_uint64, C.c_void_p]
    L.n1k_merge_partials_device.restype = C.c_int
    L.n1k_merge_partials_device.argtypes = [H, C.c_uint32, C.c_uint64, C.c_void_p]
    L.n1k_comm_unique_id.restype = C.c_int
    L.n1k_comm_unique_id.argtypes = [C.c_void_p]
    L.n1k_comm_create.restype = C.c_int
    L.n1k_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(H)]
    L.n1k_comm_destroy.restype = None
    L.n1k_comm_destroy.argtypes = [H]
    L.n1k_comm_last_error.restype = C.c_char_p
    L.n1k_comm_last_error.argtypes = [H]
    L.n1k_comm_rank.restype = C.c_int
    L.n1k_comm_rank.argtypes = [H]
    L.n1k_comm_world.restype = C.c_int
    L.n1k_comm_world.argtypes = [H]
    L.n1k_comm_max_u64.restype = C.c_int
    L.n1k_comm_max_u64.argtypes = [H, H, C.c_uint64, C.POINTER(C.c_uint64)]
    L.n1k_exchange_partials.restype = C.c_int
    L.n1k_exchange_partials.argtypes = [H, H, H, C.c_uint64, C.c_int]
    L.n1k_exchange_rows.restype = C.c_int
    L.n1k_exchange_rows.argtypes = [H, H, C.POINTER(Batch), H, C.c_uint64]
    if hasattr(L, "n1k_rows_step"):  # (absent from older builds loaded through N1K_LIB for A/B measurements)
        L.n1k_rows_step.restype = C.c_int
        L.n1k_rows_step.argtypes = [H, H, C.POINTER(Batch), H, H, C.c_uint64, C.POINTER(Result), C.POINTER(C.c_int)]
        L.n1k_gather_groups_status.restype = C.c_int
        L.n1k_gather_groups_status.argtypes = [H, H, C.POINTER(Result), C.c_int, C.POINTER(Result), C.POINTER(C.c_int)]
    if hasattr(L, "n1k_partials_step"):
        L.n1k_partials_step.restype = C.c_int
        L.n1k_partials_step.argtypes = [H, H, C.POINTER(Batch), H, H, C.c_uint64, C.c_int, C.POINTER(Result), C.POINTER(C.c_int)]
        L.n1k_failure_is_global.restype = C.c_int
        L.n1k_failure_is_global.argtypes = [H]
    if hasattr(L, "n1k_rows_step_v"):
        L.n1k_rows_step_v.restype = C.c_int
        L.n1k_rows_step_v.argtypes = [H, H, C.POINTER(Batch), H, H, C.POINTER(C.c_uint64), C.POINTER(Result), C.POINTER(C.c_int)]
        L.n1k_exchange_rows_v.restype = C.c_int
        L.n1k_exchange_rows_v.argtypes = [H, H, C.POINTER(Batch), H, C.POINTER(C.c_uint64)]
        L.n1k_exchange_sent_rows.restype = C.c_int
        L.n1k_exchange_sent_rows.argtypes = [H, H, C.POINTER(C.c_uint64)]
        L.n1k_comm_max_u64_v.restype = C.c_int
        L.n1k_comm_max_u64_v.argtypes = [H, H, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.n1k_gather_groups.restype = C.c_int
    L.n1k_gather_groups.argtypes = [H, H, C.POINTER(Result), C.POINTER(Result)]
    L.n1k_synth_columns.restype = C.c_int
    L.n1k_synth_columns.argtypes = [C.c_int, C.c_void_p, C.POINTER(SynthSpec)] + [C.c_void_p] * 7
    L.n1k_synth_documents.restype = C.c_int
    L.n1k_synth_documents.argtypes = [C.c_uint64, C.c_uint64] + [C.c_void_p] * 5 + [C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p,
                                      C.POINTER(C.c_size_t)]
    L.n1k_abi_version.restype = C.c_int
    L.n1k_device_count.restype = C.c_int
    _lib = L
    return L
