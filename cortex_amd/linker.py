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
    lut = {index.row_id(r).bytes: r for r in range(index.row_count())}
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
