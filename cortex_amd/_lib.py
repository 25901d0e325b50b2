"""ctypes binding of include/cortex_hip.h.  No fallback: a missing library is an error."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CORTEX_HIP_LIB: another build of the same library (A/B measurements of a kernel variant on one box); never a fallback
LIB_PATH = os.environ.get("CORTEX_HIP_LIB") or os.path.join(_HERE, "lib", "libcortex_hip.so")


class cx_filter(C.Structure):
    _fields_ = [
        ("has_exclude", C.c_int32), ("n_exclude", C.c_uint64), ("exclude_ids", C.c_void_p),
        ("has_kinds", C.c_int32), ("n_kinds", C.c_uint64), ("kind_codes", C.c_void_p),
        ("has_agent", C.c_int32), ("agent_code", C.c_uint32),
    ]


class cx_node_view(C.Structure):
    _fields_ = [
        ("id", C.c_uint8 * 16),
        ("kind", C.c_void_p), ("kind_len", C.c_uint64),
        ("title", C.c_void_p), ("title_len", C.c_uint64),
        ("body", C.c_void_p), ("body_len", C.c_uint64),
        ("n_tags", C.c_uint64),
        ("agent", C.c_void_p), ("agent_len", C.c_uint64),
        ("embedding", C.c_void_p), ("embedding_len", C.c_uint64),
        ("has_embedding", C.c_int32), ("importance", C.c_float),
        ("access_count", C.c_uint64),
        ("last_accessed_at_s", C.c_int64), ("last_accessed_at_ns", C.c_uint32),
        ("created_at_s", C.c_int64), ("created_at_ns", C.c_uint32),
        ("updated_at_s", C.c_int64), ("updated_at_ns", C.c_uint32),
        ("deleted", C.c_uint8),
        ("bytes_used", C.c_uint64),
    ]


class cx_decay_config(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("daily_rate", C.c_double), ("max_age_days", C.c_double), ("min_factor", C.c_double),
                ("echo_weight", C.c_double), ("echo_cap", C.c_double), ("recency_weight", C.c_float),
                ("n_by_kind", C.c_uint32), ("kind_codes", C.c_void_p), ("kind_rates", C.c_void_p)]


class cx_bulk_stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("records", "undecodable", "deleted", "no_embedding", "dim_mismatch", "indexed")]


BULK_STRICT, BULK_INCLUDE_DELETED, BULK_SET_METADATA, BULK_KEEP_ORDER, BULK_SET_STATS = 1, 2, 4, 8, 16

_P = C.c_void_p
_U64 = C.c_uint64
_U32 = C.c_uint32

# name -> (restype, argtypes): every symbol include/cortex_hip.h declares
SIGNATURES = {
    "cx_last_error": (C.c_char_p, []),
    "cx_device_count": (C.c_int, []),
    "cx_create": (_P, [_U32, C.c_int]),
    "cx_create_ex": (_P, [_U32, C.c_int, C.c_int]),
    "cx_dtype": (C.c_int, [_P]),
    "cx_destroy": (None, [_P]),
    "cx_reserve": (C.c_int, [_P, _U64]),
    "cx_upsert": (C.c_int, [_P, _P, _P, _U64]),
    "cx_upsert_batch": (C.c_int, [_P, _U64, _P, _P, _U64]),
    "cx_upsert_batch_dev": (C.c_int, [_P, _U64, _P, _P, _U64]),
    "cx_remove": (C.c_int, [_P, _P]),
    "cx_set_metadata": (C.c_int, [_P, _P, _U32, _U32]),
    "cx_set_metadata_batch": (C.c_int, [_P, _U64, _P, _P, _P]),
    "cx_intern": (_U32, [_P, C.c_char_p, _U64]),
    "cx_lookup": (_U32, [_P, C.c_char_p, _U64]),
    "cx_debug_check_result_block": (C.c_int, [_P, _P, _U64, _U64, _U64, _U64]),
    "cx_node_decode": (C.c_int, [_P, _U64, _P]),
    "cx_bulk_load_nodes": (C.c_int, [_P, _U64, _P, _P, _U32, _P]),
    "cx_set_node_stats_batch": (C.c_int, [_P, _U64, _P, _P, _P, _P, _P]),
    "cx_apply_score_decay": (C.c_float, [_P, C.c_float, C.c_float, C.c_int64, _U32, _U32, C.c_int64, _U32, _U64]),
    "cx_search_decayed": (C.c_int, [_P, _P, _U64, _U64, _U64, _P, _P, C.c_float, C.c_int64, _U32, _P, _P, _P, _P]),
    "cx_rebuild": (C.c_int, [_P]),
    "cx_save": (C.c_int, [_P, C.c_char_p]),
    "cx_load": (_P, [C.c_char_p, C.c_int]),
    "cx_load_ex": (_P, [C.c_char_p, C.c_int, C.c_int]),
    "cx_len": (_U64, [_P]),
    "cx_dimension": (_U32, [_P]),
    "cx_row_count": (_U64, [_P]),
    "cx_row_id": (C.c_int, [_P, _U64, _P]),
    "cx_rows_alive": (C.c_int, [_P, _U64, _U64, _P]),
    "cx_rows_of": (C.c_int, [_P, _U64, _P, _P]),
    "cx_search": (C.c_int, [_P, _P, _U64, _U64, _P, _P, _P, _P, _P]),
    "cx_search_threshold": (C.c_int, [_P, _P, _U64, C.c_float, _P, _U64, _P, _P, _P, _P, _P]),
    "cx_search_batch": (C.c_int, [_P, _U64, _P, _U64, _U64, _P, _P, _P, _P, _P]),
    "cx_autolink_pass_rows": (C.c_int, [_P, _U64, _P, _U64, C.c_float, _U64, _U64, _P, _P, _P, _U64, _P, _P, _P, _P, _P]),
    "cx_dedup_scan_rows": (C.c_int, [_P, C.c_float, _P, _U64, _P, _P, _P, _P, _P]),
    "cx_topk_lists_rows": (C.c_int, [_P, _U64, _P, _U64, _P, _P, _P]),
    "cx_autolink_pass_timed": (C.c_int, [_P, _U64, _P, _U64, C.c_float, _U64, _U64, _P, _P, _P, _P]),
    "cx_autolink_lists_dev": (C.c_int, [_P, _U64, _P, _U64, C.c_float, _P, _P, _P, _P, _P]),
    "cx_copy_rows_dev": (C.c_int, [_P, _U64, _U64, _P, _P]),
    "cx_search_dev": (C.c_int, [_P, _P, _U64, _P, _P, _P, _P, _P, _P]),
    "cx_search_batch_dev": (C.c_int, [_P, _U64, _P, _U64, _P, _P, _P, _P, _P, _P]),
    "cx_search_batch_streams_hint": (_U32, [_P, _U64]),
    "cx_merge_topk_dev": (C.c_int, [C.c_int, _U64, _U64, _U64, _U64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    # one index over several GPUs (sharded.cpp)
    "cx_sharded_create": (_P, [_U32, _U32, _P]),
    "cx_sharded_create_ex": (_P, [_U32, _U32, _P, C.c_int]),
    "cx_sharded_destroy": (None, [_P]),
    "cx_sharded_n_shards": (_U32, [_P]),
    "cx_sharded_shard": (_P, [_P, _U32]),
    "cx_sharded_peer_to_peer": (C.c_int, [_P]),
    "cx_sharded_upsert": (C.c_int, [_P, _P, _P, _U64]),
    "cx_sharded_upsert_batch": (C.c_int, [_P, _U64, _P, _P, _U64]),
    "cx_sharded_upsert_batch_dev": (C.c_int, [_P, _U64, _P, _P, _U64]),
    "cx_sharded_remove": (C.c_int, [_P, _P]),
    "cx_sharded_set_metadata": (C.c_int, [_P, _P, _U32, _U32]),
    "cx_sharded_intern": (_U32, [_P, C.c_char_p, _U64]),
    "cx_sharded_lookup": (_U32, [_P, C.c_char_p, _U64]),
    "cx_sharded_len": (_U64, [_P]),
    "cx_sharded_dimension": (_U32, [_P]),
    "cx_sharded_row_count": (_U64, [_P]),
    "cx_sharded_row_id": (C.c_int, [_P, _U64, _P]),
    "cx_sharded_rows_of": (C.c_int, [_P, _U64, _P, _P]),
    "cx_sharded_rebuild": (C.c_int, [_P]),
    "cx_sharded_search": (C.c_int, [_P, _P, _U64, _U64, _P, _P, _P, _P, _P]),
    "cx_sharded_search_batch": (C.c_int, [_P, _U64, _P, _U64, _U64, _P, _P, _P, _P, _P]),
    "cx_sharded_search_threshold": (C.c_int, [_P, _P, _U64, C.c_float, _P, _U64, _P, _P, _P, _P, _P]),
    "cx_sharded_autolink_pass_rows": (C.c_int, [_P, _U64, _P, _U64, C.c_float, _U64, _U64, _P, _P, _P, _U64, _P, _P, _P, _P, _P]),
    "cx_sharded_dedup_scan_rows": (C.c_int, [_P, C.c_float, _P, _U64, _P, _P, _P, _P, _P]),
    "cx_sharded_topk_lists_rows": (C.c_int, [_P, _U64, _P, _U64, _P, _P, _P]),
    "cx_sharded_set_metadata_batch": (C.c_int, [_P, _U64, _P, _P, _P]),
    "cx_sharded_bulk_load_nodes": (C.c_int, [_P, _U64, _P, _P, _U32, _P]),
    "cx_sharded_set_node_stats_batch": (C.c_int, [_P, _U64, _P, _P, _P, _P, _P]),
    "cx_sharded_search_decayed": (C.c_int, [_P, _P, _U64, _U64, _U64, _P, _P, C.c_float, C.c_int64, _U32, _P, _P, _P, _P]),
    "cx_sharded_save": (C.c_int, [_P, C.c_char_p]),
    "cx_sharded_load_ex": (_P, [C.c_char_p, _U32, _P, C.c_int]),
    "cx_profile_enable": (C.c_int, [_P, C.c_int]),
    "cx_profile_read": (C.c_int, [_P, _P, _P, C.c_int]),
    "cx_autolink_filter_profile": (C.c_int, [_P, _P]),
    "cx_device_rows": (_P, [_P]),
    # include/cortex_hip_synth.h
    "cx_synth_fill_dev": (C.c_int, [C.c_int, _P, _U64, _U64, _U64, _U64, _U64, _U64, _U32, _U32]),
    "cx_probe_read_bw": (C.c_int, [C.c_int, _U64, C.c_uint32, _P]),
    "cx_probe_mfma_tflops": (C.c_int, [C.c_int, C.c_double, _P]),
    "cx_probe_mfma_lds_tflops": (C.c_int, [C.c_int, C.c_double, C.c_int, _P]),
}

_lib = None


def load() -> C.CDLL:
    """Load libcortex_hip.so (built by cortex_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(cortex_amd has no CPU fallback)")
        # torch ships its own libamdhip64.so.7 / libhsa-runtime64; two HIP runtimes in one process
        # cannot both own the GPU ("No HIP GPUs are available").  Loading torch first makes our
        # DT_NEEDED libamdhip64.so.7 resolve to the copy torch already mapped, so the library, torch
        # tensors and RCCL share one runtime, one set of streams and one address space.  Without torch
        # (a non-Python host) the RUNPATH picks /opt/rocm/lib.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the header and the library disagree
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
