"""Host-side mirror of the reference's vector-layer interface over the C ABI.

Same names, argument meaning and error behaviour as
crates/cortex-core/src/vector/index.rs (trait VectorIndex :50-99, HnswIndex
:182-473, VectorFilter :18-47, SimilarityResult :11-15), so the parity tests
read like the reference's own.  All arithmetic happens in libcortex_hip.so on
the GPU; this module only marshals arguments.
"""
from __future__ import annotations

import ctypes as C
import uuid
from dataclasses import dataclass
from typing import Dict, Iterable, List, NamedTuple, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib

NodeId = Union[uuid.UUID, bytes]


class CortexError(Exception):
    """CortexError (error.rs:7-50)."""


class ValidationError(CortexError):
    """CortexError::Validation(String) — the only kind the vector layer raises."""


def _id16(i: NodeId) -> bytes:
    if isinstance(i, uuid.UUID):
        return i.bytes
    b = bytes(i)
    if len(b) != 16:
        raise ValidationError(f"node id must be 16 bytes, got {len(b)}")
    return b


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


class SimilarityResult(NamedTuple):
    """vector/index.rs:11-15."""
    node_id: uuid.UUID
    score: float
    distance: float


@dataclass
class VectorFilter:
    """vector/index.rs:18-47 (None = Option::None)."""
    kinds: Optional[List[str]] = None
    exclude: Optional[List[NodeId]] = None
    source_agent: Optional[str] = None

    @staticmethod
    def new() -> "VectorFilter":
        return VectorFilter()

    def with_kinds(self, kinds: Sequence[str]) -> "VectorFilter":
        self.kinds = list(kinds)
        return self

    def excluding(self, ids: Sequence[NodeId]) -> "VectorFilter":
        self.exclude = list(ids)
        return self

    def with_source_agent(self, agent: str) -> "VectorFilter":
        self.source_agent = agent
        return self


def _dtype_code(dtype: str) -> int:
    try:
        return {"f32": 0, "bf16": 1}[dtype]
    except KeyError:
        raise ValidationError(f"unknown storage dtype {dtype!r} (f32 | bf16)") from None


class HipIndex:
    """Drop-in for HnswIndex (vector/index.rs:182-473) on one MI355X.

    Exact search: inserts are visible to the next search and `rebuild()` is
    never needed for correctness."""

    def __init__(self, dimension: int, device: int = 0, dtype: str = "f32"):
        """dtype "bf16" (cx_create_ex): every inserted vector is rounded to bf16 once and kept as 2 bytes per element;
        results are the reference's for the rounded vectors."""
        self._L = _lib.load()
        self._h = self._L.cx_create_ex(dimension, device, _dtype_code(dtype))
        if not self._h:
            raise CortexError(self._err())
        self.dimension = dimension
        self.device = device
        self.dtype = dtype

    @classmethod
    def new(cls, dimension: int, device: int = 0) -> "HipIndex":
        return cls(dimension, device)

    @classmethod
    def with_metadata(cls, dimension: int, device: int = 0) -> "HipIndex":
        """vector/index.rs:214-216."""
        return cls(dimension, device)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._L.cx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ---------------------------------------------------------
    def _err(self) -> str:
        return (self._L.cx_last_error() or b"").decode(errors="replace")

    def _check(self, rc: int) -> None:
        if rc == 0:
            return
        msg = self._err()
        if rc == 1:
            raise ValidationError(msg)
        raise CortexError(msg)

    def _filter(self, f: Optional[VectorFilter]):
        if f is None:
            return None, None
        cf = _lib.cx_filter()
        keep = []
        if f.exclude is not None:
            buf = b"".join(_id16(e) for e in f.exclude)
            cb = C.create_string_buffer(buf, max(1, len(buf)))
            keep.append(cb)
            cf.has_exclude, cf.n_exclude, cf.exclude_ids = 1, len(f.exclude), C.cast(cb, C.c_void_p)
        if f.kinds is not None:
            codes = np.array([self.lookup(k) for k in f.kinds], dtype=np.uint32)
            keep.append(codes)
            cf.has_kinds, cf.n_kinds, cf.kind_codes = 1, len(f.kinds), codes.ctypes.data
        if f.source_agent is not None:
            cf.has_agent, cf.agent_code = 1, self.lookup(f.source_agent)
        return cf, keep

    def intern(self, s: str) -> int:
        """cx_intern: code of a kind / agent string, added if new — the &mut self paths (set_metadata, bulk load)."""
        b = s.encode()
        return self._L.cx_intern(self._h, b, len(b))

    def lookup(self, s: str) -> int:
        """cx_lookup: read-only (&self) — what filters use; 0 for a string no row was ever tagged with."""
        b = s.encode()
        return self._L.cx_lookup(self._h, b, len(b))

    # -- VectorIndex: mutation --------------------------------------------
    def insert(self, node_id: NodeId, embedding) -> None:
        e = _f32(embedding).reshape(-1)
        self._check(self._L.cx_upsert(self._h, _id16(node_id), e.ctypes.data, e.size))

    def insert_batch(self, ids: np.ndarray, embeddings: np.ndarray) -> None:
        """n inserts in one call: ids uint8 [n,16], embeddings f32 [n,len]."""
        e = _f32(embeddings)
        i = np.ascontiguousarray(ids, dtype=np.uint8)
        if e.ndim != 2 or i.shape != (e.shape[0], 16):
            raise ValidationError("insert_batch: ids must be [n,16] and embeddings [n,len]")
        self._check(self._L.cx_upsert_batch(self._h, e.shape[0], i.ctypes.data, e.ctypes.data, e.shape[1]))

    def insert_batch_dev(self, ids: np.ndarray, d_ptr: int, n: int, length: int) -> None:
        """Rows already resident in HBM on this device (row-major f32)."""
        i = np.ascontiguousarray(ids, dtype=np.uint8)
        if i.shape != (n, 16):
            raise ValidationError("insert_batch_dev: ids must be [n,16]")
        self._check(self._L.cx_upsert_batch_dev(self._h, n, i.ctypes.data, d_ptr, length))

    def remove(self, node_id: NodeId) -> None:
        self._check(self._L.cx_remove(self._h, _id16(node_id)))

    def set_metadata(self, node_id: NodeId, kind: str, source_agent: str) -> None:
        self._check(self._L.cx_set_metadata(self._h, _id16(node_id), self.intern(kind), self.intern(source_agent)))

    def set_metadata_batch(self, ids, kinds, source_agents) -> None:
        """cx_set_metadata_batch: set_metadata for many ids with one device copy."""
        arr = ids if isinstance(ids, np.ndarray) else np.frombuffer(b"".join(_id16(i) for i in ids), dtype=np.uint8)
        arr = np.ascontiguousarray(arr, dtype=np.uint8).reshape(-1, 16)
        kc = np.asarray([self.intern(k) for k in kinds], np.uint32)
        ac = np.asarray([self.intern(a) for a in source_agents], np.uint32)
        if not (len(arr) == len(kc) == len(ac)):
            raise ValidationError("ids, kinds and source_agents must have the same length")
        self._check(self._L.cx_set_metadata_batch(self._h, len(arr), arr.ctypes.data, kc.ctypes.data, ac.ctypes.data))

    def bulk_load_nodes(self, records, strict: bool = False, include_deleted: bool = False,
                        set_metadata: bool = False, keep_order: bool = False, set_stats: bool = False) -> dict:
        """The start-up loop of serve.rs:105-123 / api.rs:56-70 over raw bincode `Node` records (the values
        of the reference's nodes table): decode, drop deleted / embedding-less nodes, insert newest first.
        Returns the counters of cx_bulk_stats.  strict=True is Cortex::open's behaviour (a wrong-length
        embedding is an error), False is the server's (the node is skipped)."""
        recs = [bytes(r) for r in records]
        offs = np.zeros(len(recs) + 1, np.uint64)
        if recs:
            offs[1:] = np.cumsum([len(r) for r in recs], dtype=np.uint64)
        blob = np.frombuffer(b"".join(recs) or b"\0", dtype=np.uint8)
        flags = ((_lib.BULK_STRICT if strict else 0) | (_lib.BULK_INCLUDE_DELETED if include_deleted else 0)
                 | (_lib.BULK_SET_METADATA if set_metadata else 0) | (_lib.BULK_KEEP_ORDER if keep_order else 0)
                 | (_lib.BULK_SET_STATS if set_stats else 0))
        st = _lib.cx_bulk_stats()
        rc = self._L.cx_bulk_load_nodes(self._h, len(recs), blob.ctypes.data, offs.ctypes.data, flags, C.byref(st))
        self._check(rc)
        return {n: int(getattr(st, n)) for n, _ in st._fields_}

    def set_node_stats(self, ids, kinds, last_accessed_at, access_counts) -> None:
        """cx_set_node_stats_batch: the node fields apply_score_decay reads (scoring.rs:84-114) for many ids;
        last_accessed_at = sequence of (seconds, nanoseconds) since the epoch."""
        arr = ids if isinstance(ids, np.ndarray) else np.frombuffer(b"".join(_id16(i) for i in ids), dtype=np.uint8)
        arr = np.ascontiguousarray(arr, dtype=np.uint8).reshape(-1, 16)
        kc = np.asarray([self.intern(k) for k in kinds], np.uint32)
        ls = np.asarray([t[0] for t in last_accessed_at], np.int64)
        lns = np.asarray([t[1] for t in last_accessed_at], np.uint32)
        ac = np.asarray(access_counts, np.uint64)
        if not (len(arr) == len(kc) == len(ls) == len(ac)):
            raise ValidationError("ids, kinds, last_accessed_at and access_counts must have the same length")
        self._check(self._L.cx_set_node_stats_batch(self._h, len(arr), arr.ctypes.data, kc.ctypes.data, ls.ctypes.data,
                                                    lns.ctypes.data, ac.ctypes.data))

    def search_decayed(self, query, limit: int, config, recency_bias: Optional[float] = None, now=None,
                       filter: Optional["VectorFilter"] = None, candidate_limit: Optional[int] = None):
        """The HTTP search handler's sequence (routes.rs:889-947) in one call: candidates, apply_score_decay,
        stable re-rank, truncate.  -> [(node_id, score, raw_score)], len <= limit.  config: scoring.ScoreDecayConfig."""
        from . import scoring
        rb = config.recency_weight if recency_bias is None else recency_bias
        cl = scoring.http_candidate_limit(limit, config, rb) if candidate_limit is None else candidate_limit
        now = scoring.now_utc() if now is None else now
        q = _f32(query).reshape(-1)
        c, keep = config._c(self.lookup)
        cap = max(1, limit)
        ids = np.zeros((cap, 16), np.uint8)
        sc = np.zeros(cap, np.float32)
        raw = np.zeros(cap, np.float32)
        n = C.c_uint64(0)
        f, fkeep = self._filter(filter)
        self._check(self._L.cx_search_decayed(self._h, q.ctypes.data, len(q), limit, cl, C.byref(f) if f else None, C.byref(c), rb, now[0], now[1],
                                              ids.ctypes.data, sc.ctypes.data, raw.ctypes.data, C.byref(n)))
        return [(uuid.UUID(bytes=ids[j].tobytes()), float(sc[j]), float(raw[j])) for j in range(n.value)]

    def rebuild(self) -> None:
        self._check(self._L.cx_rebuild(self._h))

    def reserve(self, rows: int) -> None:
        self._check(self._L.cx_reserve(self._h, rows))

    def save(self, path) -> None:
        """vector/index.rs:437-445 — the reference's bincode file format."""
        self._check(self._L.cx_save(self._h, str(path).encode()))

    @classmethod
    def load(cls, path, device: int = 0, dtype: str = "f32") -> "HipIndex":
        """vector/index.rs:447-473."""
        L = _lib.load()
        h = L.cx_load_ex(str(path).encode(), device, _dtype_code(dtype))
        if not h:
            msg = (L.cx_last_error() or b"").decode(errors="replace")
            raise ValidationError(msg)
        self = cls.__new__(cls)
        self._L, self._h, self.device, self.dtype = L, h, device, dtype
        self.dimension = int(L.cx_dimension(h))
        return self

    # -- VectorIndex: queries ----------------------------------------------
    def __len__(self) -> int:
        return int(self._L.cx_len(self._h))

    def len(self) -> int:
        return len(self)

    def is_empty(self) -> bool:
        return len(self) == 0

    def row_count(self) -> int:
        return int(self._L.cx_row_count(self._h))

    def row_id(self, row: int) -> uuid.UUID:
        out = (C.c_uint8 * 16)()
        self._check(self._L.cx_row_id(self._h, row, out))
        return uuid.UUID(bytes=bytes(out))

    def device_rows_ptr(self) -> int:
        return int(self._L.cx_device_rows(self._h) or 0)

    def search_arrays(self, query, k: int, filter: Optional[VectorFilter] = None
                      ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """search() returning (ids uint8 [n,16], scores f32 [n], distances f32 [n])."""
        q = _f32(query).reshape(-1)
        kk = max(1, min(int(k), max(1, self.row_count())))
        ids = np.zeros((kk, 16), dtype=np.uint8)
        scores = np.zeros(kk, dtype=np.float32)
        dists = np.zeros(kk, dtype=np.float32)
        n = C.c_uint64(0)
        cf, keep = self._filter(filter)
        self._check(self._L.cx_search(self._h, q.ctypes.data, q.size, int(k), C.byref(cf) if cf else None,
                                      ids.ctypes.data, scores.ctypes.data, dists.ctypes.data, C.byref(n)))
        return ids[:n.value], scores[:n.value], dists[:n.value]

    def search(self, query, k: int, filter: Optional[VectorFilter] = None) -> List[SimilarityResult]:
        ids, s, d = self.search_arrays(query, k, filter)
        return [SimilarityResult(uuid.UUID(bytes=ids[i].tobytes()), float(s[i]), float(d[i])) for i in range(len(s))]

    def search_threshold_arrays(self, query, threshold: float, filter: Optional[VectorFilter] = None
                                ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        q = _f32(query).reshape(-1)
        cf, keep = self._filter(filter)
        cap = 1024
        while True:
            ids = np.zeros((cap, 16), dtype=np.uint8)
            scores = np.zeros(cap, dtype=np.float32)
            dists = np.zeros(cap, dtype=np.float32)
            n, need = C.c_uint64(0), C.c_uint64(0)
            rc = self._L.cx_search_threshold(self._h, q.ctypes.data, q.size, float(threshold),
                                             C.byref(cf) if cf else None, cap, ids.ctypes.data,
                                             scores.ctypes.data, dists.ctypes.data, C.byref(n), C.byref(need))
            if rc == 4:  # CX_ERR_CAPACITY: count-then-fill
                cap = int(need.value)
                continue
            self._check(rc)
            return ids[:n.value], scores[:n.value], dists[:n.value]

    def search_threshold(self, query, threshold: float, filter: Optional[VectorFilter] = None
                         ) -> List[SimilarityResult]:
        ids, s, d = self.search_threshold_arrays(query, threshold, filter)
        return [SimilarityResult(uuid.UUID(bytes=ids[i].tobytes()), float(s[i]), float(d[i])) for i in range(len(s))]

    def search_batch_arrays(self, queries, k: int, filter: Optional[VectorFilter] = None):
        """queries f32 [nq,len] -> (ids [nq,k,16], scores [nq,k], distances [nq,k], counts [nq])."""
        qs = _f32(queries)
        if qs.ndim != 2:
            raise ValidationError("search_batch: queries must be [nq,len]")
        nq = qs.shape[0]
        kk = max(1, int(k))
        ids = np.zeros((nq, kk, 16), dtype=np.uint8)
        scores = np.zeros((nq, kk), dtype=np.float32)
        dists = np.zeros((nq, kk), dtype=np.float32)
        counts = np.zeros(nq, dtype=np.uint64)
        cf, keep = self._filter(filter)
        self._check(self._L.cx_search_batch(self._h, nq, qs.ctypes.data, qs.shape[1], int(k),
                                            C.byref(cf) if cf else None, ids.ctypes.data, scores.ctypes.data,
                                            dists.ctypes.data, counts.ctypes.data))
        return ids, scores, dists, counts

    def search_batch(self, queries: Iterable[Tuple[NodeId, Sequence[float]]], k: int,
                     filter: Optional[VectorFilter] = None) -> Dict[uuid.UUID, List[SimilarityResult]]:
        """vector/index.rs:390-410: &[(NodeId, Embedding)] -> HashMap<NodeId, Vec<SimilarityResult>>."""
        queries = list(queries)
        if not queries:
            return {}
        qs = np.stack([_f32(e).reshape(-1) for _, e in queries])
        ids, s, d, cnt = self.search_batch_arrays(qs, k, filter)
        out: Dict[uuid.UUID, List[SimilarityResult]] = {}
        for qi, (qid, _) in enumerate(queries):
            key = qid if isinstance(qid, uuid.UUID) else uuid.UUID(bytes=_id16(qid))
            out[key] = [SimilarityResult(uuid.UUID(bytes=ids[qi, j].tobytes()), float(s[qi, j]), float(d[qi, j]))
                        for j in range(int(cnt[qi]))]
        return out

    # -- the auto-linker's similarity pass, batched ---------------------------
    @staticmethod
    def _existing_csr(existing, n_scan: int):
        """(offsets u64 [n_scan+1], to_rows u32) -> contiguous arrays for the ABI; None passes through."""
        if existing is None:
            return None, None
        eo = np.ascontiguousarray(existing[0], dtype=np.uint64)
        et = np.ascontiguousarray(existing[1], dtype=np.uint32)
        if eo.size != n_scan + 1 or int(eo[-1]) != et.size:
            raise ValidationError("existing: offsets must have n_scan + 1 entries and end at len(to_rows)")
        return eo, et

    def autolink_pass_rows(self, scan_rows, topk: int, threshold: float, max_edges_per_node: int,
                           deleted: Optional[np.ndarray] = None, existing=None,
                           max_edges_per_cycle: Optional[int] = None):
        """cx_autolink_pass_rows: (from_rows u32, to_rows u32, weights f32), scan order then score order.
        scan_rows=None scans every row.  existing = (offsets [n_scan+1], to_rows) CSR of the related_to edges the
        scanned nodes already have (auto_linker.rs:226-231); max_edges_per_cycle: :284-287, None = no truncation."""
        sr = None if scan_rows is None else np.ascontiguousarray(scan_rows, dtype=np.uint32)
        n_scan = self.row_count() if sr is None else sr.size
        dl = None if deleted is None else np.ascontiguousarray(deleted, dtype=np.uint8)
        if dl is not None and dl.size != self.row_count():
            raise ValidationError("deleted must have one flag per row")
        eo, et = self._existing_csr(existing, n_scan)
        cyc = 0xFFFFFFFFFFFFFFFF if max_edges_per_cycle is None else int(max_edges_per_cycle)
        cap = max(1024, n_scan * 4)
        while True:
            fr, to = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
            w = np.zeros(cap, np.float32)
            n, need = C.c_uint64(0), C.c_uint64(0)
            rc = self._L.cx_autolink_pass_rows(self._h, n_scan, sr.ctypes.data if sr is not None else None, int(topk),
                                               float(threshold), int(max_edges_per_node), cyc,
                                               dl.ctypes.data if dl is not None else None,
                                               eo.ctypes.data if eo is not None else None,
                                               et.ctypes.data if et is not None else None, cap, fr.ctypes.data,
                                               to.ctypes.data, w.ctypes.data, C.byref(n), C.byref(need))
            if rc == 4 and need.value > cap:
                cap = int(need.value)
                continue
            self._check(rc)
            return fr[:n.value], to[:n.value], w[:n.value]

    def dedup_scan_rows(self, dedup_threshold: float, deleted: Optional[np.ndarray] = None):
        """cx_dedup_scan_rows: (a_rows, b_rows, similarity) — DedupScanner::scan's pairs (dedup.rs:65-127)."""
        dl = None if deleted is None else np.ascontiguousarray(deleted, dtype=np.uint8)
        cap = max(1024, self.row_count())
        while True:
            a, b = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
            s = np.zeros(cap, np.float32)
            n, need = C.c_uint64(0), C.c_uint64(0)
            rc = self._L.cx_dedup_scan_rows(self._h, float(dedup_threshold), dl.ctypes.data if dl is not None else None,
                                            cap, a.ctypes.data, b.ctypes.data, s.ctypes.data, C.byref(n), C.byref(need))
            if rc == 4 and need.value > cap:
                cap = int(need.value)
                continue
            self._check(rc)
            return a[:n.value], b[:n.value], s[:n.value]

    def rows_alive(self, row_lo: int = 0, n: Optional[int] = None) -> np.ndarray:
        """cx_rows_alive: bool per row of [row_lo, row_lo + n): False = removed (a tombstone until rebuild)."""
        n = self.row_count() - row_lo if n is None else n
        out = np.zeros(max(1, n), np.uint8)
        self._check(self._L.cx_rows_alive(self._h, row_lo, n, out.ctypes.data))
        return out[:n].astype(bool)

    def rows_of(self, ids) -> np.ndarray:
        """cx_rows_of: insertion rows of ids (u8 [n,16] or a sequence of 16-byte ids); UINT32_MAX = not indexed."""
        arr = ids if isinstance(ids, np.ndarray) else np.frombuffer(b"".join(_id16(i) for i in ids), dtype=np.uint8)
        arr = np.ascontiguousarray(arr, dtype=np.uint8).reshape(-1, 16)
        out = np.zeros(len(arr), np.uint32)
        self._check(self._L.cx_rows_of(self._h, len(arr), arr.ctypes.data, out.ctypes.data))
        return out

    def topk_lists_rows(self, topk: int, scan_rows=None):
        """cx_topk_lists_rows: (rows [n_scan, topk] u32, scores [n_scan, topk] f32, counts [n_scan]) — the ordered
        top-k neighbour list of every scanned row, self included (auto_linker.rs:221 for a whole batch of nodes)."""
        sr = None if scan_rows is None else np.ascontiguousarray(scan_rows, dtype=np.uint32)
        n_scan = self.row_count() if sr is None else sr.size
        rows = np.zeros((n_scan, int(topk)), np.uint32)
        scores = np.zeros((n_scan, int(topk)), np.float32)
        counts = np.zeros(n_scan, np.uint32)
        self._check(self._L.cx_topk_lists_rows(self._h, n_scan, sr.ctypes.data if sr is not None else None, int(topk),
                                               rows.ctypes.data, scores.ctypes.data, counts.ctypes.data))
        return rows, scores, counts

    def autolink_pass_timed(self, topk: int, threshold: float, max_edges_per_node: int, scan_rows=None,
                            existing=None, max_edges_per_cycle: Optional[int] = None):
        """(n_edges, [shadow_ms, filter_ms, rescore_ms, rules_ms]) with the edges left in HBM."""
        sr = None if scan_rows is None else np.ascontiguousarray(scan_rows, dtype=np.uint32)
        n_scan = self.row_count() if sr is None else sr.size
        eo, et = self._existing_csr(existing, n_scan)
        cyc = 0xFFFFFFFFFFFFFFFF if max_edges_per_cycle is None else int(max_edges_per_cycle)
        ne = C.c_uint64(0)
        ph = (C.c_double * 4)()
        self._check(self._L.cx_autolink_pass_timed(self._h, n_scan, sr.ctypes.data if sr is not None else None,
                                                   int(topk), float(threshold), int(max_edges_per_node), cyc,
                                                   eo.ctypes.data if eo is not None else None,
                                                   et.ctypes.data if et is not None else None,
                                                   C.byref(ne), ph))
        return ne.value, list(ph)

    def autolink_filter_profile(self) -> dict:
        """cx_autolink_filter_profile: the MFMA filter GEMM of the last timed pass (kernel ms, executed flops, tiles, kernel, clock)."""
        out = (C.c_double * 5)()
        self._check(self._L.cx_autolink_filter_profile(self._h, out))
        kern = {0: "cx::pair_filter256_kernel (retired in round 4)", 1: "cx::pair_filter_p_kernel", 2: "cx::pair_filter_kernel / pair_filter_stream_kernel"}[int(out[3])]
        return {"kernel_ms": out[0], "executed_flops": out[1], "tiles": int(out[2]), "kernel": kern, "shader_clock_ghz": out[4]}

    def autolink_lists_dev(self, d_queries: int, nq: int, topk: int, threshold: float, d_rows: int, d_scores: int,
                           d_dists: int, d_counts: int, stream: int = 0) -> None:
        """cx_autolink_lists_dev: ordered neighbour lists of nq external vectors (HBM) against this shard."""
        self._check(self._L.cx_autolink_lists_dev(self._h, nq, d_queries, int(topk), float(threshold), d_rows, d_scores,
                                                  d_dists, d_counts, stream))

    def copy_rows_dev(self, row_lo: int, n: int, d_dst: int, stream: int = 0) -> None:
        self._check(self._L.cx_copy_rows_dev(self._h, row_lo, n, d_dst, stream))

    # -- measurement ---------------------------------------------------------
    def profile_enable(self, on: bool = True) -> None:
        self._check(self._L.cx_profile_enable(self._h, 1 if on else 0))

    def profile_read(self, reset: bool = True) -> Tuple[float, int]:
        """(sum of scan-kernel durations in ms, number of launches) since the last reset."""
        ms, n = C.c_double(0.0), C.c_uint64(0)
        self._check(self._L.cx_profile_read(self._h, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value

    # -- HBM-resident variants ---------------------------------------------
    def search_batch_dev(self, d_queries: int, nq: int, k: int, d_rows: int, d_scores: int, d_dists: int,
                         d_counts: int, stream: int = 0, filter: Optional[VectorFilter] = None) -> None:
        cf, keep = self._filter(filter)
        self._check(self._L.cx_search_batch_dev(self._h, nq, d_queries, int(k), C.byref(cf) if cf else None,
                                                d_rows, d_scores, d_dists, d_counts, stream))


    def search_batch_streams_hint(self, nq: int) -> int:
        """cx_search_batch_streams_hint: HIP streams worth rotating over for a stream of batches of nq queries on this index."""
        return int(self._L.cx_search_batch_streams_hint(self._h, int(nq)))


class _ShardedAbi:
    """Presents the cx_sharded_* entry points under the names of their single-index counterparts, so that
    ShardedHipIndex IS HipIndex with a different handle: every method that has a sharded form runs unchanged; one
    that has none fails loudly instead of reaching a shard behind the index's back."""
    _HAVE = {"upsert", "upsert_batch", "upsert_batch_dev", "remove", "set_metadata", "set_metadata_batch", "intern", "lookup",
             "len", "dimension", "row_count", "row_id", "rows_of", "rebuild", "search", "search_batch", "search_threshold",
             "autolink_pass_rows", "dedup_scan_rows", "topk_lists_rows", "bulk_load_nodes", "set_node_stats_batch",
             "search_decayed", "save"}

    def __init__(self, L):
        self._L = L

    def __getattr__(self, name):
        if name.startswith("cx_") and name[3:] in self._HAVE:
            return getattr(self._L, "cx_sharded_" + name[3:])
        if name in ("cx_last_error", "cx_device_count") or name.startswith("cx_sharded_"):
            return getattr(self._L, name)
        raise ValidationError(f"{name} has no sharded form: call it on a shard (ShardedHipIndex.shard(i)) or on a HipIndex")


class ShardedHipIndex(HipIndex):
    """One index over several GPUs of the node behind the same interface (include/cortex_hip.h: cx_sharded).
    devices: one entry per shard; a device may repeat.  Rows in the linker interfaces (autolink_pass_rows,
    dedup_scan_rows, rows_of, row_id, row_count) are GLOBAL rows = insertion sequence numbers, i.e. exactly the rows of
    a single HipIndex that saw the same calls."""

    def __init__(self, dimension: int, devices: Sequence[int], dtype: str = "f32"):
        L = _lib.load()
        self._L = _ShardedAbi(L)
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        self._h = L.cx_sharded_create_ex(dimension, len(devices), devs, _dtype_code(dtype))
        self.dtype = dtype
        if not self._h:
            raise CortexError(self._err())
        self.dimension = dimension
        self.devices = list(devices)
        self.device = self.devices[0]

    @classmethod
    def new(cls, dimension: int, devices: Sequence[int] = (0,)) -> "ShardedHipIndex":
        return cls(dimension, devices)

    @classmethod
    def load(cls, path, devices: Sequence[int] = (0,), dtype: str = "f32") -> "ShardedHipIndex":
        """VectorIndex::load (index.rs:447-473) into a sharded handle: the file of a single index or of a sharded one
        (same layout), vectors placed like insert_batch would."""
        L = _lib.load()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = L.cx_sharded_load_ex(str(path).encode(), len(devices), devs, _dtype_code(dtype))
        if not h:
            raise CortexError(L.cx_last_error().decode("utf-8", "replace"))
        self = cls.__new__(cls)
        self._L = _ShardedAbi(L)
        self._h = h
        self.dtype = dtype
        self.dimension = int(L.cx_sharded_dimension(h))
        self.devices = list(devices)
        self.device = self.devices[0]
        return self

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._L.cx_sharded_destroy(self._h)
            self._h = None

    @property
    def n_shards(self) -> int:
        return int(self._L.cx_sharded_n_shards(self._h))

    @property
    def peer_to_peer(self) -> bool:
        return bool(self._L.cx_sharded_peer_to_peer(self._h))

    def shard_len(self, i: int) -> int:
        """rows held by shard i (live ones)"""
        L = _lib.load()
        return int(L.cx_len(L.cx_sharded_shard(self._h, i)))
