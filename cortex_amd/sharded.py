"""Row-range sharded kNN across the GPUs of one node.

One process per GPU.  Each rank owns a contiguous row range of the corpus in
its own HipIndex, runs the same scan on its shard, and the only exchange step
is one RCCL all-gather (torch.distributed backend "nccl") of the packed
per-shard partial top-k lists — (3*nq*k + nq) 4-byte words per rank, KB-scale,
latency-bound over xGMI — followed by a k-way merge kernel on every rank.
The reference has no counterpart (single process, no collectives); the result
is what `HnswIndex::search` (vector/index.rs:325-374, exact path) would return
on the concatenated corpus, ties resolved by global row.

The local search and the merge are injectable so the collective plumbing can
be exercised on CPU (gloo) in tests; the defaults are the HIP paths.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, Optional, Sequence

import os

import numpy as np
import torch
import torch.distributed as dist


def packed_words(nq: int, k: int) -> int:
    """4-byte words of one rank's chunk: rows[nq*k] | scores[nq*k] | dists[nq*k] | counts[nq], padded to 16 B."""
    w = 3 * nq * k + nq
    return (w + 3) // 4 * 4


class ShardedKnn:
    def __init__(self, rank: int, world: int, row_bases: Sequence[int], nq: int, k: int,
                 device: torch.device, local_fn: Callable[[int, int, "ShardedKnn"], None],
                 merge_fn: Optional[Callable[["ShardedKnn"], None]] = None, group=None):
        """row_bases[p] = global index of shard p's first row (ascending).
        local_fn(query_ptr_or_tensor, nq, self) must fill self.local (packed chunk) on the current stream.
        merge_fn(self) must fold self.gathered into self.out_* ; None = the HIP merge kernel."""
        assert len(row_bases) == world
        self.rank, self.world, self.k, self.nq = rank, world, k, nq
        self.row_bases = np.asarray(row_bases, dtype=np.uint64)
        self.device = device
        self.group = group
        self.words = packed_words(nq, k)
        # two sets of exchange buffers: submit() pipelines the all-gather of batch i under the scan of batch i+1
        self._locals = [torch.zeros(self.words, dtype=torch.int32, device=device) for _ in range(2)]
        self._gathered = [torch.zeros(world * self.words, dtype=torch.int32, device=device) for _ in range(2)]
        self.local, self.gathered = self._locals[0], self._gathered[0]
        self._pending = None   # (work handle, buffer index) of the all-gather still in flight
        self._seq = 0
        self.out_rows = torch.zeros((nq, k), dtype=torch.int64, device=device)
        self.out_scores = torch.zeros((nq, k), dtype=torch.float32, device=device)
        self.out_dists = torch.zeros((nq, k), dtype=torch.float32, device=device)
        self.out_counts = torch.zeros(nq, dtype=torch.int32, device=device)
        self.local_fn = local_fn
        self.merge_fn = merge_fn or _hip_merge
        # On a GPU submit() runs a stream of batches on `depth` HIP streams in rotation, each with its own exchange and result
        # buffers: batch i's scan, all-gather and merge are enqueued on stream i % depth, so the fixed costs of one batch's pass
        # (its prologue, its first bounds, its tail) and its exchange run under the next batches' streaming — measured with
        # concurrent readers of one shard (profiles/r03/tuning.md 8.7): +24 % batches per second at four in flight.
        self._streams = None
        # (batches only: single-query scans are one kernel at the stream's rate each — several in flight gain nothing and their
        # HIP-event durations would span each other)
        # (how many: the shard's own answer for calls of this size — cx_search_batch_streams_hint: 4 for 64-query passes, 2 for the
        # 128-query passes of row widths up to 512, which lose when two of them are in each other's way; tuning.md 1.10)
        hint = getattr(local_fn, "streams_hint", None)
        default_depth = (hint(nq) if hint is not None else 4) if nq >= 3 else 0
        depth = int(os.environ.get("CX_SHARDED_STREAMS", str(default_depth))) if device.type == "cuda" else 0
        if depth >= 2:
            self._streams = [torch.cuda.Stream(device=device) for _ in range(depth)]
            self._locals = [torch.zeros(self.words, dtype=torch.int32, device=device) for _ in range(depth)]
            self._gathered = [torch.zeros(world * self.words, dtype=torch.int32, device=device) for _ in range(depth)]
            self._outs = [(torch.zeros((nq, k), dtype=torch.int64, device=device), torch.zeros((nq, k), dtype=torch.float32, device=device),
                           torch.zeros((nq, k), dtype=torch.float32, device=device), torch.zeros(nq, dtype=torch.int32, device=device))
                          for _ in range(depth)]
            self.local, self.gathered = self._locals[0], self._gathered[0]

    # views into a packed chunk (base = tensor of `words` int32)
    def chunk_views(self, base: torch.Tensor):
        n = self.nq * self.k
        rows = base[0:n].view(self.nq, self.k)
        scores = base[n:2 * n].view(torch.float32).view(self.nq, self.k)
        dists = base[2 * n:3 * n].view(torch.float32).view(self.nq, self.k)
        counts = base[3 * n:3 * n + self.nq]
        return rows, scores, dists, counts

    def search(self, queries) -> None:
        """One batch of nq queries: local scan, all-gather, merge.  Asynchronous on the current stream;
        results land in out_rows / out_scores / out_dists / out_counts."""
        self.local_fn(queries, self.nq, self)
        if self.world == 1 and self.merge_fn is _hip_merge:
            return  # single shard: the local list is the answer (read it with chunk_views(self.local))
        if self.world > 1:
            dist.all_gather_into_tensor(self.gathered, self.local, group=self.group)
        else:
            self.gathered.copy_(self.local)
        self.merge_fn(self)


    # -- pipelined stream of batches ------------------------------------------------------------
    def submit(self, queries) -> None:
        """Like search(), for a stream of batches: the all-gather of this batch is left in flight (async_op)
        and its merge is enqueued by the NEXT submit()/flush(), after that call's local scan — so the
        exchange latency hides under the next scan.  Results of a batch are complete after the following
        submit() or flush(); out_* always hold the most recently merged batch."""
        if self._streams is not None:
            i = self._seq % len(self._streams)
            st = self._streams[i]
            st.wait_stream(torch.cuda.current_stream(self.device))   # the queries were produced on the caller's stream
            with torch.cuda.stream(st):
                self.local, self.gathered = self._locals[i], self._gathered[i]
                self.out_rows, self.out_scores, self.out_dists, self.out_counts = self._outs[i]
                self.local_fn(queries, self.nq, self)
                if self.world > 1:
                    work = dist.all_gather_into_tensor(self._gathered[i], self._locals[i], group=self.group, async_op=True)
                    work.wait()   # this stream waits for the collective; the host does not block, the other streams run on
                    self.merge_fn(self)
                elif self.merge_fn is not _hip_merge:
                    self._gathered[i].copy_(self._locals[i])
                    self.merge_fn(self)
            self._seq += 1
            return
        i = self._seq & 1
        self.local, self.gathered = self._locals[i], self._gathered[i]
        self.local_fn(queries, self.nq, self)
        prev = self._pending
        if self.world > 1:
            work = dist.all_gather_into_tensor(self._gathered[i], self._locals[i], group=self.group, async_op=True)
        else:
            work = None
            if self.merge_fn is not _hip_merge:
                self._gathered[i].copy_(self._locals[i])
        self._pending = (work, i)
        self._seq += 1
        if prev is not None:
            self._finish(prev)

    def flush(self) -> None:
        if self._streams is not None:   # the caller's stream waits for every batch in flight
            cur = torch.cuda.current_stream(self.device)
            for st in self._streams:
                cur.wait_stream(st)
            return
        if self._pending is not None:
            self._finish(self._pending)
            self._pending = None

    def _finish(self, pending) -> None:
        work, i = pending
        if work is not None:
            work.wait()   # the current stream waits for the collective; the host does not block
        if self.world == 1 and self.merge_fn is _hip_merge:
            return
        keep = self.local, self.gathered
        self.local, self.gathered = self._locals[i], self._gathered[i]
        self.merge_fn(self)
        self.local, self.gathered = keep


def hip_local_fn(index) -> Callable:
    """local_fn for a cortex_amd.HipIndex shard: queries = device pointer (int) to nq*dim f32."""
    def fn(d_queries: int, nq: int, s: ShardedKnn) -> None:
        n = s.nq * s.k
        base = s.local.data_ptr()
        stream = torch.cuda.current_stream(s.device).cuda_stream
        index.search_batch_dev(d_queries, nq, s.k, base, base + 4 * n, base + 8 * n, base + 12 * n, stream)
    fn.streams_hint = index.search_batch_streams_hint
    return fn


def _hip_merge(s: ShardedKnn) -> None:
    from . import _lib
    L = _lib.load()
    n = s.nq * s.k
    base = s.gathered.data_ptr()
    stream = torch.cuda.current_stream(s.device).cuda_stream
    rc = L.cx_merge_topk_dev(s.device.index or 0, s.world, s.nq, s.k, s.words,
                             s.row_bases.ctypes.data, base, base + 4 * n, base + 8 * n, base + 12 * n,
                             s.out_rows.data_ptr(), s.out_scores.data_ptr(), s.out_dists.data_ptr(),
                             s.out_counts.data_ptr(), stream)
    if rc:
        raise RuntimeError((L.cx_last_error() or b"").decode())


class ShardedAutolink:
    """The auto-linker's all-pairs similarity pass over row-range shards: every rank owns a shard;
    the scanned rows travel in blocks (the owner broadcasts a block of its f32 rows), every rank produces the
    block's ordered neighbour lists against its own shard (cx_autolink_lists_dev: bf16 MFMA filter + exact
    rescore), the lists are all-gathered and merged exactly like partial top-k lists (ShardedKnn), and the owner
    walks the merged lists with the reference's rules (linker/auto_linker.rs:233-264, rules.rs:42-62).
    Dedup needs no collective at all: each rank's threshold pairs are final (not built here).

    lists_fn(queries, nq, knn) fills knn.local for a block; rows_fn(row_lo, n, tensor) copies own rows into the
    block buffer.  Both are injectable (CPU gloo tests); hip_lists_fn / hip_rows_fn are the product paths."""

    def __init__(self, rank: int, world: int, shard_rows: Sequence[int], dim: int, topk: int, device: torch.device,
                 lists_fn: Callable, rows_fn: Callable, merge_fn=None, group=None, block: int = 2048,
                 alive_fn: Optional[Callable] = None):
        self.rank, self.world, self.dim, self.topk, self.block = rank, world, dim, topk, block
        self.shard_rows = [int(x) for x in shard_rows]
        self.bases = np.concatenate([[0], np.cumsum(self.shard_rows)[:-1]]).astype(np.int64)
        self.device, self.group = device, group
        self.rows_fn = rows_fn
        self.alive_fn = alive_fn   # alive_fn(row_lo, n) -> bool[n] over OWN rows: removed rows are not scanned (auto_linker.rs:217-218)
        self.knn = ShardedKnn(rank, world, self.bases.tolist(), block, topk, device, lists_fn, merge_fn=merge_fn, group=group)
        self.buf = torch.zeros((block, dim), dtype=torch.float32, device=device)

    def run(self, threshold: float, max_edges_per_node: int, deleted: Optional[np.ndarray] = None, existing=None,
            max_edges_per_cycle: Optional[int] = None):
        """Edges proposed for the rows this rank owns: (from_global i64, to_global i64, weight f32), scan order then
        score order — AutoLinker::run_cycle's walk (auto_linker.rs:215-264) in global rows, as the fused pass of one
        index runs it (linker.walk_similarity_lists):
        deleted: optional flags over GLOBAL rows (storage tombstones, quirk Q2);
        existing: optional (offsets u64 [own_rows + 1], to_global) CSR of the related_to edges each OWN row already has
        (:226-231): dropped without counting towards max_edges_per_node (:249-258);
        max_edges_per_cycle: :284-287 `take(max_edges_per_cycle)` over the cycle's proposals in scan order = global row
        order: a rank keeps what the ranks before it left of the budget (one all_gather of the ranks' edge counts)."""
        from .linker import walk_similarity_lists
        thr = np.float32(threshold)
        out_f, out_t, out_w = [], [], []
        for src in range(self.world):
            n_src = self.shard_rows[src]
            for lo in range(0, n_src, self.block):
                m = min(self.block, n_src - lo)
                if self.rank == src:
                    self.buf.zero_()
                    self.rows_fn(lo, m, self.buf)
                if self.world > 1:
                    dist.broadcast(self.buf, src, group=self.group)
                self.knn.live = m
                single = self.world == 1 and self.knn.merge_fn is _hip_merge
                self.knn.search(self.buf)
                if self.rank != src:
                    continue
                if single:   # one shard: the local lists are the answer
                    if self.device.type == "cuda":
                        torch.cuda.synchronize(self.device)
                    r, sc, _, cnt = self.knn.chunk_views(self.knn.local)
                    rows = (r[:m].cpu().numpy().astype(np.int64) & 0xFFFFFFFF)
                    scores, counts = sc[:m].cpu().numpy(), cnt[:m].cpu().numpy()
                else:
                    if self.device.type == "cuda":
                        torch.cuda.synchronize(self.device)
                    rows = self.knn.out_rows[:m].cpu().numpy()
                    scores, counts = self.knn.out_scores[:m].cpu().numpy(), self.knn.out_counts[:m].cpu().numpy()
                self_g = self.bases[src] + lo + np.arange(m, dtype=np.int64)
                counts = np.asarray(counts).copy()
                if self.alive_fn is not None:   # a row removed from the index has no embedding: it proposes nothing (:217-218)
                    counts[~np.asarray(self.alive_fn(lo, m), dtype=bool)] = 0
                ex = None
                if existing is not None:
                    eo = np.asarray(existing[0], dtype=np.int64)
                    ex = (eo[lo:lo + m + 1] - eo[lo], np.asarray(existing[1])[eo[lo]:eo[lo + m]])
                f, t, w = walk_similarity_lists(self_g, rows, scores, counts, thr, max_edges_per_node, deleted, ex)
                out_f.append(f); out_t.append(t); out_w.append(w)
        cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
        f, t, w = cat(out_f, np.int64), cat(out_t, np.int64), cat(out_w, np.float32)
        if max_edges_per_cycle is not None:
            before = 0
            if self.world > 1:
                mine = torch.tensor([len(f)], dtype=torch.int64, device=self.device if self.device.type == "cuda" else "cpu")
                every = [torch.zeros_like(mine) for _ in range(self.world)]
                dist.all_gather(every, mine, group=self.group)
                before = int(sum(int(x.item()) for x in every[:self.rank]))
            keep = max(0, min(len(f), int(max_edges_per_cycle) - before))
            f, t, w = f[:keep], t[:keep], w[:keep]
        return f, t, w


def hip_lists_fn(index, threshold: float) -> Callable:
    """lists_fn for a cortex_amd.HipIndex shard: queries = torch f32 [block, dim] on the device."""
    def fn(queries: torch.Tensor, nq: int, s: ShardedKnn) -> None:
        n = s.nq * s.k
        base = s.local.data_ptr()
        stream = torch.cuda.current_stream(s.device).cuda_stream
        index.autolink_lists_dev(queries.data_ptr(), nq, s.k, threshold, base, base + 4 * n, base + 8 * n, base + 12 * n, stream)
    return fn


def hip_alive_fn(index) -> Callable:
    """alive_fn for a cortex_amd.HipIndex shard."""
    return lambda row_lo, n: index.rows_alive(row_lo, n)


def hip_rows_fn(index) -> Callable:
    def fn(row_lo: int, n: int, buf: torch.Tensor) -> None:
        index.copy_rows_dev(row_lo, n, buf.data_ptr(), torch.cuda.current_stream(buf.device).cuda_stream)
    return fn


def reference_merge(s: ShardedKnn) -> None:
    """Plain torch/numpy statement of the merge (score desc, global row asc, NaN last) — used by the
    CPU gloo tests and as the checker of the HIP merge kernel in the GPU tests."""
    g = s.gathered.cpu()
    cand = [[] for _ in range(s.nq)]
    for p in range(s.world):
        rows, scores, dists, counts = s.chunk_views(g[p * s.words:(p + 1) * s.words])
        for q in range(s.nq):
            for j in range(int(counts[q])):
                sc = float(scores[q, j])
                grow = int(s.row_bases[p]) + (int(rows[q, j]) & 0xFFFFFFFF)
                cand[q].append((np.isnan(sc), -sc if not np.isnan(sc) else 0.0, grow, sc, float(dists[q, j])))
    out_rows = torch.zeros((s.nq, s.k), dtype=torch.int64)
    out_scores = torch.zeros((s.nq, s.k), dtype=torch.float32)
    out_dists = torch.zeros((s.nq, s.k), dtype=torch.float32)
    out_counts = torch.zeros(s.nq, dtype=torch.int32)
    for q in range(s.nq):
        best = sorted(cand[q])[:s.k]
        out_counts[q] = len(best)
        for j, (_, _, grow, sc, di) in enumerate(best):
            out_rows[q, j], out_scores[q, j], out_dists[q, j] = grow, sc, di
    s.out_rows.copy_(out_rows)
    s.out_scores.copy_(out_scores)
    s.out_dists.copy_(out_dists)
    s.out_counts.copy_(out_counts)
