"""Regenerates tests/golden/*.npz: inputs + expected outputs for the exact cosine path.

Expected values come from oracle/np_twin.py — an independent numpy restatement
of crates/cortex-core/src/vector/index.rs:169-179, :254-256, :259-294 (the
reference is Rust and cannot run in this image; it holds no numeric golden
vectors of its own).  tests/test_oracle_golden.py checks the C
oracle against these files bit for bit; the GPU parity tests check the HIP
path against them within the stated tolerance.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import np_twin as T  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
F = np.float32


def clustered(rng, n, d, n_centres, sigma, normalise=True):
    c = rng.standard_normal((n_centres, d)).astype(F)
    c /= np.linalg.norm(c, axis=1, keepdims=True).astype(F)
    who = rng.integers(0, n_centres, n)
    x = c[who] + (sigma / np.sqrt(d)).astype(F) * rng.standard_normal((n, d)).astype(F)
    if normalise:
        x /= np.linalg.norm(x, axis=1, keepdims=True).astype(F)
    return x.astype(F)


def expected(rows, queries, k):
    er = np.zeros((len(queries), k), np.int32)
    es = np.zeros((len(queries), k), F)
    ed = np.zeros((len(queries), k), F)
    for i, q in enumerate(queries):
        r, s, d = T.brute_force(q, rows, k)
        er[i, :len(r)], es[i, :len(r)], ed[i, :len(r)] = r, s, d
    return er, es, ed


def main():
    # 1. unit rows, d=384 (BASELINE config 1/2 shape), k=10
    rng = np.random.default_rng(20260313)
    rows = clustered(rng, 128, 384, 6, F(0.42))
    qs = clustered(rng, 6, 384, 6, F(0.42))
    qs[0] = rows[17]  # an exact self-match
    er, es, ed = expected(rows, qs, 10)
    np.savez(os.path.join(HERE, "unit_384.npz"), rows=rows, queries=qs, k=10, exp_rows=er, exp_scores=es, exp_dists=ed)

    # 2. un-normalised rows (norms in [0.5, 2]) to pin the division, d=768, k=5
    rng = np.random.default_rng(20260314)
    rows = clustered(rng, 64, 768, 4, F(0.6))
    rows *= rng.uniform(0.5, 2.0, size=(64, 1)).astype(F)
    qs = clustered(rng, 4, 768, 4, F(0.6)) * F(1.7)
    er, es, ed = expected(rows, qs, 5)
    np.savez(os.path.join(HERE, "unnorm_768.npz"), rows=rows, queries=qs, k=5, exp_rows=er, exp_scores=es, exp_dists=ed)

    # 3. ties: exact duplicates, anti-parallel rows (score clamps to 0), one zero row (NaN)
    rng = np.random.default_rng(20260315)
    base = clustered(rng, 24, 32, 3, F(0.5))
    rows = np.concatenate([base, base[[3, 3, 7]], -base[[0, 1]], np.zeros((1, 32), F), base[[3]]]).astype(F)
    qs = np.stack([base[3], base[0], rng.standard_normal(32).astype(F)])
    er, es, ed = expected(rows, qs, len(rows))
    np.savez(os.path.join(HERE, "ties_32.npz"), rows=rows, queries=qs, k=len(rows), exp_rows=er, exp_scores=es,
             exp_dists=ed)

    # 4. the reference's own 3-d vectors (vector/index.rs:484-728)
    rows = np.array([[1, 0, 0], [0.9, 0.1, 0], [0, 1, 0], [-1, 0, 0], [0, 0, 1]], F)
    qs = np.array([[1, 0, 0], [0, 1, 0]], F)
    er, es, ed = expected(rows, qs, 5)
    np.savez(os.path.join(HERE, "ref_3d.npz"), rows=rows, queries=qs, k=5, exp_rows=er, exp_scores=es, exp_dists=ed)
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
