"""Host-side mirror of the auto-linker's similarity pass over the C ABI.

Mirrors, for the part of `AutoLinker::run_cycle` that consumes similarity scores
(crates/cortex-core/src/linker/auto_linker.rs:215-264) and `DedupScanner::scan`
(linker/dedup.rs:65-127): same inputs (scanned nodes, SimilarityConfig thresholds,
max_edges_per_node), same outputs (ProposedEdge with relation "related_to",
weight = score, provenance AutoSimilarity{score} — linker/rules.rs:42-62;
DuplicatePair{node_a, node_b, similarity}).  Cursor handling, storage writes,
decay and the string-heuristic rules stay with the reference (out of scope).
"""
from __future__ import annotations

import uuid
from dataclasses import dataclass, field
from typing import Iterable, List, Optional, Sequence

import numpy as np

from .config import SimilarityConfig
from .index import HipIndex, NodeId, _id16

AUTO_LINK_TOPK = 100  # hard-coded in the reference (auto_linker.rs:221); SimilarityConfig.auto_link_k is dead


@dataclass
class ProposedEdge:
    """linker/rules.rs:7-14 for SimilarityLinkRule."""
    from_id: uuid.UUID
    to_id: uuid.UUID
    weight: float
    relation: str = "related_to"
    provenance: dict = field(default_factory=dict)


@dataclass
class DuplicatePair:
    """linker/dedup.rs:24-31 (the suggestion is decided by the reference's merge logic, out of scope)."""
    node_a: uuid.UUID
    node_b: uuid.UUID
    similarity: float


def _rows_of(index: HipIndex, ids: Iterable[NodeId]) -> np.ndarray:
    if hasattr(index, "rows_of"):
        rows = index.rows_of(list(ids))
        if np.any(rows == np.uint32(0xFFFFFFFF)):
            raise KeyError("id not in the index")
        return rows
    lut = {index.row_id(r).bytes: r for r in range(index.row_count())}   # duck-typed indexes (tests)
    return np.array([lut[_id16(i)] for i in ids], dtype=np.uint32)


def _deleted_flags(index: HipIndex, deleted_ids: Optional[Sequence[NodeId]]) -> Optional[np.ndarray]:
    if not deleted_ids:
        return None
    flags = np.zeros(index.row_count(), dtype=np.uint8)
    flags[_rows_of(index, deleted_ids)] = 1
    return flags


def autolink_similarity_edges(index: HipIndex, scan_ids: Optional[Sequence[NodeId]], config: SimilarityConfig,
                              max_edges_per_node: int = 50, deleted_ids: Optional[Sequence[NodeId]] = None,
                              topk: int = AUTO_LINK_TOPK) -> List[ProposedEdge]:
    """The edges `run_cycle` proposes from SimilarityLinkRule for the scanned nodes (scan order, then
    score order), before its existing-edge and per-cycle filters.  scan_ids=None scans every node."""
    config.validate()
    scan_rows = None if scan_ids is None else _rows_of(index, scan_ids)
    fr, to, w = index.autolink_pass_rows(scan_rows, topk, config.auto_link_threshold, max_edges_per_node,
                                         _deleted_flags(index, deleted_ids))
    ids = {}

    def rid(r: int) -> uuid.UUID:
        if r not in ids:
            ids[r] = index.row_id(int(r))
        return ids[r]

    return [ProposedEdge(rid(a), rid(b), float(s), provenance={"AutoSimilarity": {"score": float(s)}})
            for a, b, s in zip(fr, to, w)]


def dedup_scan(index: HipIndex, config: SimilarityConfig,
               deleted_ids: Optional[Sequence[NodeId]] = None) -> List[DuplicatePair]:
    a, b, s = index.dedup_scan_rows(config.dedup_threshold, _deleted_flags(index, deleted_ids))
    return [DuplicatePair(index.row_id(int(x)), index.row_id(int(y)), float(z)) for x, y, z in zip(a, b, s)]


def neighbour_lists(index: HipIndex, scan_ids: Optional[Sequence[NodeId]] = None, topk: int = AUTO_LINK_TOPK):
    """`vector_index.search(&embedding, 100, None)` (auto_linker.rs:221) for every scanned node in one call:
    (scan_rows, rows [n, topk], scores [n, topk], counts [n]); lists are ordered best first and contain the node
    itself (the walk below skips it, like the reference)."""
    scan_rows = None if scan_ids is None else _rows_of(index, scan_ids)
    rows, scores, counts = index.topk_lists_rows(topk, scan_rows)
    if scan_rows is None:
        scan_rows = np.arange(index.row_count(), dtype=np.uint32)
    return scan_rows, rows, scores, counts


def autolink_walk(index: HipIndex, scan_ids: Optional[Sequence[NodeId]], rules, max_edges_per_node: int = 50,
                  deleted_ids: Optional[Sequence[NodeId]] = None, existing=None, topk: int = AUTO_LINK_TOPK):
    """The reference's per-node loop (auto_linker.rs:215-264) over the GPU's ordered neighbour lists, for ANY set
    of link rules (SURVEY §8 a14': with the legacy structural rules on, every rule's edges count towards the
    per-node cap, so the walk needs the ordered top-100 lists, not a thresholded pass).

    rules: callables (node_row, neighbour_row, score) -> iterable of (relation, weight); applied in order for
    each neighbour (apply_link_rules).  existing: optional callable node_row -> set of (to_row, relation)
    (the pre-loaded `existing_set`, :226-230).  Returns [(from_row, to_row, relation, weight)] in the order
    run_cycle proposes them.  The cap is tested after each NEIGHBOUR (:259-262), so a node can exceed it by the
    edges its last neighbour added — exactly as the reference does."""
    scan_rows, rows, scores, counts = neighbour_lists(index, scan_ids, topk)
    deleted = _deleted_flags(index, deleted_ids)
    out = []
    for p, node in enumerate(scan_rows):
        have = existing(int(node)) if existing is not None else ()
        n_edges = 0
        for c in range(int(counts[p])):
            nb, score = int(rows[p, c]), float(scores[p, c])
            if nb == int(node):
                continue                                   # skip self (:233-236)
            if deleted is not None and deleted[nb]:
                continue                                   # storage-deleted neighbour (:238-242)
            for rule in rules:
                for relation, weight in rule(int(node), nb, score):
                    if (nb, relation) not in have:         # existing-edge filter (:252-256)
                        n_edges += 1
                        out.append((int(node), nb, relation, float(weight)))
            if n_edges >= max_edges_per_node:              # per-node limit (:259-262)
                break
    return out


def walk_similarity_lists(node_rows, rows, scores, counts, threshold, max_edges_per_node: int, deleted=None, existing=None):
    """auto_linker.rs:233-264 with SimilarityLinkRule (rules.rs:42-62) as the only rule, over MANY nodes' ordered neighbour
    lists at once (numpy; the per-neighbour Python loop of autolink_walk is for arbitrary rules).  Same walk as
    link_rules_kernel (allpairs.hip):
      skip self (:235-237) and storage-deleted neighbours (:240-243) — they do not reach the cap test;
      score >= threshold -> an edge, unless the node already has it (existing_set, :226-231, :249-258: dropped WITHOUT counting);
      after every neighbour that was not skipped, stop once max_edges_per_node edges were proposed (:259-262 — tested after the
      push, so a cap of 0 still lets the first neighbour's edge through).
    node_rows [m] (the rows the lists belong to), rows / scores [m, k], counts [m]; deleted: flags indexed by row;
    existing: (offsets u64 [m + 1], to_rows) CSR over these m nodes.  Returns (from, to, weight), node order then list order."""
    node_rows = np.asarray(node_rows, dtype=np.int64)
    rows = np.asarray(rows).astype(np.int64)
    scores = np.asarray(scores, dtype=np.float32)
    counts = np.asarray(counts).astype(np.int64)
    m, k = rows.shape
    valid = np.arange(k)[None, :] < counts[:, None]
    safe = np.where(valid, rows, 0)
    considered = valid & (rows != node_rows[:, None])
    if deleted is not None:
        considered &= ~np.asarray(deleted)[safe].astype(bool)
    emit = considered & (scores >= np.float32(threshold))
    if existing is not None:
        eo = np.asarray(existing[0], dtype=np.int64)
        et = np.asarray(existing[1], dtype=np.int64)
        if et.size:
            owner = np.repeat(np.arange(m, dtype=np.int64), np.diff(eo))
            have = np.unique((owner << 32) | et)
            keys = (np.arange(m, dtype=np.int64)[:, None] << 32) | safe
            pos = np.searchsorted(have, keys)
            hit = (pos < have.size) & (have[np.minimum(pos, have.size - 1)] == keys)
            emit &= ~hit
    cum = np.cumsum(emit, axis=1)
    stop = considered & (cum >= max_edges_per_node)               # the walk ends after the first of these
    first_stop = np.where(stop.any(axis=1), stop.argmax(axis=1), k)
    emit &= np.arange(k)[None, :] <= first_stop[:, None]
    ii, jj = np.nonzero(emit)
    return node_rows[ii], rows[ii, jj], scores[ii, jj]


def similarity_rule(config: SimilarityConfig):
    """SimilarityLinkRule::evaluate (linker/rules.rs:42-62) as an autolink_walk rule."""
    thr = np.float32(config.auto_link_threshold)

    def rule(node: int, neighbour: int, score: float):
        return [("related_to", score)] if np.float32(score) >= thr else []
    return rule
