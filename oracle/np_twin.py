"""Independent numpy restatement of the reference's exact distance path.

TEST INFRASTRUCTURE ONLY.  Written separately from cortex_oracle.c so that the
two restatements check each other (the reference itself cannot run here and
holds no numeric golden vectors for this path).  Also the generator of the
committed fixtures under tests/golden/ (tests/golden/make_golden.py).

Sequential f32 accumulation is spelled np.cumsum(..., dtype=float32)[-1]:
np.sum / np.dot use pairwise or BLAS orders, which is not what
`iter().zip().map(|(a, b)| a * b).sum::<f32>()` (vector/index.rs:172) does.
"""
from __future__ import annotations

import numpy as np

F = np.float32


def _seq_sum(x: np.ndarray) -> np.float32:
    if x.size == 0:
        return F(0.0)
    return np.cumsum(x.astype(F), dtype=F)[-1]


def distance(a: np.ndarray, b: np.ndarray) -> np.float32:
    """EmbeddingPoint::distance, vector/index.rs:169-179."""
    a = a.astype(F)
    b = b.astype(F)
    with np.errstate(invalid="ignore", divide="ignore", over="ignore", under="ignore"):
        dot = _seq_sum(a * b)
        na = np.sqrt(_seq_sum(a * a), dtype=F)
        nb = np.sqrt(_seq_sum(b * b), dtype=F)
        sim = F(dot) / F(na * nb)
        return F(F(1.0) - sim)


def distances(q: np.ndarray, rows: np.ndarray) -> np.ndarray:
    """distance(q, row) for every row, sequential sums along the feature axis."""
    q = q.astype(F)
    rows = rows.astype(F)
    with np.errstate(invalid="ignore", divide="ignore", over="ignore", under="ignore"):
        dot = np.cumsum(rows * q[None, :], axis=1, dtype=F)[:, -1] if rows.shape[1] else np.zeros(len(rows), F)
        nr = np.sqrt(np.cumsum(rows * rows, axis=1, dtype=F)[:, -1], dtype=F) if rows.shape[1] else np.zeros(len(rows), F)
        nq = np.sqrt(_seq_sum(q * q), dtype=F)
        sim = (dot / (nq * nr).astype(F)).astype(F)
        return (F(1.0) - sim).astype(F)


def distance_to_similarity(d: np.ndarray) -> np.ndarray:
    """vector/index.rs:254-256; np.clip propagates NaN like f32::clamp."""
    return np.clip((F(1.0) - np.asarray(d, dtype=F)).astype(F), F(0.0), F(1.0)).astype(F)


def brute_force(q: np.ndarray, rows: np.ndarray, k: int, keep: np.ndarray | None = None):
    """vector/index.rs:259-294 with the declared order (score desc, row asc, NaN last).

    Returns (row_indices, scores, distances)."""
    d = distances(q, rows)
    s = distance_to_similarity(d)
    idx = np.arange(len(rows))
    if keep is not None:
        idx = idx[keep]
    key = np.where(np.isnan(s[idx]), -np.inf, s[idx].astype(np.float64))
    order = np.argsort(-key, kind="stable")
    sel = idx[order][:k]
    return sel, s[sel], d[sel]
