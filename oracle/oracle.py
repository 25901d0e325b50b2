"""ctypes loader for the CPU restatement (oracle/cortex_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under cortex_amd/ may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcortex_oracle.so")


def build(force: bool = False) -> str:
    """Compile the C restatement with the flags in oracle/Makefile."""
    srcs = [os.path.join(_HERE, f) for f in ("cortex_oracle.c", "cortex_synth.c", "cortex_hnsw.c", "cortex_oracle.h", "Makefile")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Result(C.Structure):
    _fields_ = [("node_id", C.c_uint8 * 16), ("score", C.c_float), ("distance", C.c_float), ("row", C.c_uint32)]


class _Filter(C.Structure):
    _fields_ = [("has_kinds", C.c_int), ("n_kinds", C.c_size_t), ("kinds", C.POINTER(C.c_char_p)),
                ("has_exclude", C.c_int), ("n_exclude", C.c_size_t), ("exclude", C.c_void_p),
                ("has_agent", C.c_int), ("source_agent", C.c_char_p)]


class _Edge(C.Structure):
    _fields_ = [("from_row", C.c_uint32), ("to_row", C.c_uint32), ("weight", C.c_float)]


class _Config(C.Structure):
    _fields_ = [("auto_link_threshold", C.c_float), ("dedup_threshold", C.c_float),
                ("contradiction_threshold", C.c_float), ("auto_link_k", C.c_size_t)]


RESULT_DTYPE = np.dtype([("node_id", "u1", 16), ("score", "<f4"), ("distance", "<f4"), ("row", "<u4")])
EDGE_DTYPE = np.dtype([("from_row", "<u4"), ("to_row", "<u4"), ("weight", "<f4")])

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.cxo_distance.restype = C.c_float
        L.cxo_distance.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.cxo_distance_to_similarity.restype = C.c_float
        L.cxo_distance_to_similarity.argtypes = [C.c_float]
        L.cxo_index_new.restype = C.c_void_p
        L.cxo_index_new.argtypes = [C.c_size_t]
        L.cxo_index_free.argtypes = [C.c_void_p]
        L.cxo_insert.restype = C.c_int
        L.cxo_insert.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.cxo_insert_batch.restype = C.c_int
        L.cxo_insert_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
        L.cxo_remove.restype = C.c_int
        L.cxo_remove.argtypes = [C.c_void_p, C.c_void_p]
        L.cxo_set_metadata.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p]
        L.cxo_len.restype = C.c_size_t
        L.cxo_len.argtypes = [C.c_void_p]
        L.cxo_row_count.restype = C.c_size_t
        L.cxo_row_count.argtypes = [C.c_void_p]
        L.cxo_search.restype = C.c_size_t
        L.cxo_search.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.cxo_search_threshold.restype = C.c_size_t
        L.cxo_search_threshold.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
        L.cxo_search_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_void_p]
        L.cxo_config_default.argtypes = [C.c_void_p]
        L.cxo_config_clamp.restype = C.c_float
        L.cxo_config_clamp.argtypes = [C.c_float]
        L.cxo_config_validate.restype = C.c_int
        L.cxo_config_validate.argtypes = [C.c_void_p]
        L.cxo_autolink_pass.restype = C.c_size_t
        L.cxo_autolink_pass.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_float, C.c_size_t, C.c_size_t,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.cxo_dedup_scan.restype = C.c_size_t
        L.cxo_dedup_scan.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.cxo_last_error.restype = C.c_char_p
        L.cxs_row.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_size_t,
                              C.c_uint32, C.c_void_p]
        L.cxs_fill.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64,
                               C.c_size_t, C.c_uint32, C.c_void_p]
        L.cxs_centre.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, C.c_void_p]
        L.cxo_hnsw_build.restype = C.c_void_p
        L.cxo_hnsw_build.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_uint64]
        L.cxo_hnsw_search.restype = C.c_size_t
        L.cxo_hnsw_search.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        L.cxo_hnsw_build_mt.restype = C.c_void_p
        L.cxo_hnsw_build_mt.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int]
        L.cxo_hnsw_search_batch.restype = None
        L.cxo_hnsw_search_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cxo_hnsw_dist_evals.restype = C.c_uint64
        L.cxo_hnsw_dist_evals.argtypes = [C.c_void_p]
        L.cxo_hnsw_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _id_bytes(i) -> bytes:
    if isinstance(i, (bytes, bytearray)):
        assert len(i) == 16
        return bytes(i)
    if hasattr(i, "bytes"):  # uuid.UUID
        return i.bytes
    return np.asarray(i, dtype=np.uint8).tobytes()


@dataclass
class Filter:
    """VectorFilter (vector/index.rs:18-47): None = Option::None."""
    kinds: Optional[Sequence[str]] = None
    exclude: Optional[Sequence[bytes]] = None
    source_agent: Optional[str] = None

    def _c(self):
        f = _Filter()
        keep = []
        if self.kinds is not None:
            arr = (C.c_char_p * max(1, len(self.kinds)))(*[k.encode() for k in self.kinds])
            keep.append(arr)
            f.has_kinds, f.n_kinds, f.kinds = 1, len(self.kinds), arr
        if self.exclude is not None:
            buf = b"".join(_id_bytes(e) for e in self.exclude)
            cb = C.create_string_buffer(buf, max(1, len(buf)))
            keep.append(cb)
            f.has_exclude, f.n_exclude, f.exclude = 1, len(self.exclude), C.cast(cb, C.c_void_p)
        if self.source_agent is not None:
            f.has_agent, f.source_agent = 1, self.source_agent.encode()
        return f, keep


class OracleError(Exception):
    """CortexError::Validation(String) (error.rs)."""


class OracleIndex:
    """HnswIndex on its exact path (vector/index.rs:182-473)."""

    def __init__(self, dimension: int):
        self._L = lib()
        self._h = self._L.cxo_index_new(dimension)
        self.dimension = dimension

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.cxo_index_free(self._h)
            self._h = None

    def insert(self, node_id, embedding) -> None:
        e = _f32(embedding)
        if self._L.cxo_insert(self._h, _id_bytes(node_id), e.ctypes.data, e.size):
            raise OracleError(self._L.cxo_last_error().decode())

    def insert_batch(self, ids: np.ndarray, embs: np.ndarray) -> None:
        embs = _f32(embs)
        ids = np.ascontiguousarray(ids, dtype=np.uint8)
        assert ids.shape == (embs.shape[0], 16)
        if self._L.cxo_insert_batch(self._h, embs.shape[0], ids.ctypes.data, embs.ctypes.data, embs.shape[1]):
            raise OracleError(self._L.cxo_last_error().decode())

    def remove(self, node_id) -> None:
        self._L.cxo_remove(self._h, _id_bytes(node_id))

    def set_metadata(self, node_id, kind: str, source_agent: str) -> None:
        self._L.cxo_set_metadata(self._h, _id_bytes(node_id), kind.encode(), source_agent.encode())

    def __len__(self) -> int:
        return self._L.cxo_len(self._h)

    def is_empty(self) -> bool:
        return len(self) == 0

    def row_count(self) -> int:
        return self._L.cxo_row_count(self._h)

    def rebuild(self) -> None:
        """vector/index.rs:416-435 builds the HNSW graph; the exact path has nothing to build."""

    def search(self, query, k: int, filter: Optional[Filter] = None) -> np.ndarray:
        q = _f32(query)
        out = np.zeros(max(1, min(k, len(self))), dtype=RESULT_DTYPE)
        f, keep = filter._c() if filter else (None, None)
        n = self._L.cxo_search(self._h, q.ctypes.data, k, C.byref(f) if filter else None, out.ctypes.data)
        return out[:n]

    def search_threshold(self, query, threshold: float, filter: Optional[Filter] = None) -> np.ndarray:
        q = _f32(query)
        out = np.zeros(max(1, len(self)), dtype=RESULT_DTYPE)
        f, keep = filter._c() if filter else (None, None)
        n = self._L.cxo_search_threshold(self._h, q.ctypes.data, threshold, C.byref(f) if filter else None,
                                         out.ctypes.data)
        return out[:n]

    def search_batch(self, queries, k: int, filter: Optional[Filter] = None, n_threads: int = 1) -> List[np.ndarray]:
        qs = _f32(queries)
        nq = qs.shape[0]
        k_eff = max(1, min(k, max(1, len(self))))
        out = np.zeros((nq, k_eff), dtype=RESULT_DTYPE)
        counts = np.zeros(nq, dtype=np.uintp)
        f, keep = filter._c() if filter else (None, None)
        self._L.cxo_search_batch(self._h, nq, qs.ctypes.data, min(k, k_eff), C.byref(f) if filter else None, n_threads,
                                 out.ctypes.data, counts.ctypes.data)
        return [out[i, :counts[i]] for i in range(nq)]

    def autolink_pass(self, scan_rows, topk: int, threshold: float, max_edges_per_node: int,
                      deleted: Optional[np.ndarray] = None, n_threads: int = 1, existing=None,
                      max_edges_per_cycle: Optional[int] = None) -> np.ndarray:
        """existing: (offsets u64 [n_scan+1], to_rows u32) CSR of the related_to edges each scanned node already
        has (auto_linker.rs:226-231); max_edges_per_cycle: :284-287 (None = no truncation)."""
        rows = np.ascontiguousarray(scan_rows, dtype=np.uint32)
        cap = max(1, rows.size * max(max_edges_per_node, 1))
        out = np.zeros(cap, dtype=EDGE_DTYPE)
        need = C.c_size_t(0)
        d = np.ascontiguousarray(deleted, dtype=np.uint8) if deleted is not None else None
        eo = et = None
        if existing is not None:
            eo = np.ascontiguousarray(existing[0], dtype=np.uint64)
            et = np.ascontiguousarray(existing[1], dtype=np.uint32)
            assert eo.size == rows.size + 1 and int(eo[-1]) == et.size
        cyc = (1 << 63) if max_edges_per_cycle is None else int(max_edges_per_cycle)
        n = self._L.cxo_autolink_pass(self._h, rows.size, rows.ctypes.data, topk, threshold, max_edges_per_node, cyc,
                                      d.ctypes.data if d is not None else None,
                                      eo.ctypes.data if eo is not None else None,
                                      et.ctypes.data if et is not None else None, n_threads,
                                      out.ctypes.data, cap, C.byref(need))
        return out[:n]

    def dedup_scan(self, dedup_threshold: float, deleted: Optional[np.ndarray] = None) -> np.ndarray:
        cap = 1024
        d = np.ascontiguousarray(deleted, dtype=np.uint8) if deleted is not None else None
        while True:
            out = np.zeros(cap, dtype=EDGE_DTYPE)
            need = C.c_size_t(0)
            n = self._L.cxo_dedup_scan(self._h, dedup_threshold, d.ctypes.data if d is not None else None,
                                       out.ctypes.data, cap, C.byref(need))
            if need.value <= cap:
                return out[:n]
            cap = need.value


def distance(a, b) -> float:
    a, b = _f32(a), _f32(b)
    return float(lib().cxo_distance(a.ctypes.data, b.ctypes.data, a.size))


def distance_to_similarity(d: float) -> float:
    return float(lib().cxo_distance_to_similarity(d))


@dataclass
class SimilarityConfig:
    """vector/config.rs:3-87."""
    auto_link_threshold: float = 0.75
    dedup_threshold: float = 0.92
    contradiction_threshold: float = 0.80
    auto_link_k: int = 20

    @staticmethod
    def default() -> "SimilarityConfig":
        c = _Config()
        lib().cxo_config_default(C.byref(c))
        return SimilarityConfig(c.auto_link_threshold, c.dedup_threshold, c.contradiction_threshold, c.auto_link_k)

    def validate(self) -> None:
        c = _Config(self.auto_link_threshold, self.dedup_threshold, self.contradiction_threshold, self.auto_link_k)
        if lib().cxo_config_validate(C.byref(c)):
            raise OracleError(lib().cxo_last_error().decode())


def clamp_threshold(t: float) -> float:
    return float(lib().cxo_config_clamp(t))


# ---------------------------------------------------------------- synthetic

SEED_CORPUS = 20260313
SEED_QUERIES = 20260314
SEED_DUP = 20260315


def synth_rows(n_total: int, d: int, row_lo: int = 0, n_rows: Optional[int] = None, *, flags: int = 1,
               seed_rows: int = SEED_CORPUS, seed_centres: int = SEED_CORPUS, seed_dup: int = SEED_DUP) -> np.ndarray:
    """Rows [row_lo, row_lo+n_rows) of the n_total-row synthetic corpus."""
    n_rows = n_total - row_lo if n_rows is None else n_rows
    out = np.empty((n_rows, d), dtype=np.float32)
    lib().cxs_fill(seed_centres, seed_rows, seed_dup, max(1, n_total // 50), row_lo, n_rows, d, flags,
                   out.ctypes.data)
    return out


def synth_queries(n_total: int, d: int, nq: int, *, seed_centres: int = SEED_CORPUS) -> np.ndarray:
    """Held-out queries: same centres as the n_total-row corpus, fresh noise, no duplicates."""
    return synth_rows(n_total, d, 0, nq, flags=0, seed_rows=SEED_QUERIES, seed_centres=seed_centres)


class HnswBaseline:
    """CPU HNSW restatement (oracle/cortex_hnsw.c): a REPORTED BASELINE for the reference's approximate path
    (instant-distance 0.6.1, not in /root/reference).  Parity unpinned — never used as a checker."""

    def __init__(self, rows: np.ndarray, M: int = 32, M0: int = 64, ef_construction: int = 100, seed: int = 1, n_threads: int = 1):
        """n_threads > 1: the concurrent build (the reference builds with rayon, index.rs:430)."""
        self._L = lib()
        self.rows = _f32(rows)
        if n_threads > 1:
            self._h = self._L.cxo_hnsw_build_mt(self.rows.ctypes.data, self.rows.shape[0], self.rows.shape[1], M, M0,
                                                ef_construction, seed, n_threads)
        else:
            self._h = self._L.cxo_hnsw_build(self.rows.ctypes.data, self.rows.shape[0], self.rows.shape[1], M, M0,
                                             ef_construction, seed)

    def search_batch(self, queries, k: int, ef_search: int = 100, n_threads: int = 1):
        """-> (rows [nq][k] u32, dist [nq][k] f32, counts [nq])"""
        qs = _f32(queries)
        nq = qs.shape[0]
        rows = np.zeros((nq, max(1, k)), dtype=np.uint32)
        dist = np.zeros((nq, max(1, k)), dtype=np.float32)
        counts = np.zeros(nq, dtype=np.uintp)
        self._L.cxo_hnsw_search_batch(self._h, qs.ctypes.data, nq, k, ef_search, n_threads, rows.ctypes.data, dist.ctypes.data, counts.ctypes.data)
        return rows, dist, counts

    def search(self, query, k: int, ef_search: int = 100):
        q = _f32(query)
        rows = np.zeros(max(1, k), dtype=np.uint32)
        dist = np.zeros(max(1, k), dtype=np.float32)
        n = self._L.cxo_hnsw_search(self._h, q.ctypes.data, k, ef_search, rows.ctypes.data, dist.ctypes.data)
        return rows[:n], dist[:n]

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.cxo_hnsw_free(self._h)
            self._h = None
