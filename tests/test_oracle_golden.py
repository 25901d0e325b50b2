"""C oracle vs the committed golden fixtures (numpy-twin generated) — bit for bit — and
vs the numpy twin on fresh seeded inputs."""
import glob
import os

import numpy as np
import pytest

from conftest import ids_for

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_matches_golden(oracle, path):
    g = np.load(path)
    rows, qs, k = g["rows"], g["queries"], int(g["k"])
    ix = oracle.OracleIndex(rows.shape[1])
    ix.insert_batch(ids_for(len(rows)), rows)
    for i, q in enumerate(qs):
        r = ix.search(q, k)
        n = len(r)
        assert np.array_equal(r["row"], g["exp_rows"][i][:n])
        assert np.array_equal(r["score"], g["exp_scores"][i][:n], equal_nan=True)
        assert np.array_equal(r["distance"], g["exp_dists"][i][:n], equal_nan=True)


@pytest.mark.parametrize("n,d,seed", [(300, 64, 1), (100, 384, 2), (50, 1000, 3), (17, 3, 4)])
def test_oracle_matches_numpy_twin(oracle, n, d, seed):
    from oracle import np_twin as T
    rng = np.random.default_rng(seed)
    rows = rng.standard_normal((n, d)).astype(np.float32)
    ix = oracle.OracleIndex(d)
    ix.insert_batch(ids_for(n), rows)
    for _ in range(4):
        q = rng.standard_normal(d).astype(np.float32)
        r = ix.search(q, 10)
        sel, s, dist = T.brute_force(q, rows, 10)
        assert np.array_equal(r["row"], sel)
        assert np.array_equal(r["score"], s)
        assert np.array_equal(r["distance"], dist)
        assert oracle.distance(q, rows[0]) == T.distance(q, rows[0])


def test_upsert_keeps_row_and_remove_reinserts_at_end(oracle):
    ix = oracle.OracleIndex(2)
    ids = ids_for(3)
    ix.insert(ids[0].tobytes(), [1, 0])
    ix.insert(ids[1].tobytes(), [1, 0])
    ix.insert(ids[0].tobytes(), [1, 0])  # upsert: stays row 0
    assert len(ix) == 2
    r = ix.search([1, 0], 2)
    assert list(r["row"]) == [0, 1]
    ix.remove(ids[0].tobytes())
    ix.insert(ids[0].tobytes(), [1, 0])  # new row 2, after row 1
    r = ix.search([1, 0], 2)
    assert list(r["row"]) == [1, 2]


def test_synth_generator_properties(oracle):
    x = oracle.synth_rows(4000, 96)
    assert np.allclose(np.linalg.norm(x, axis=1), 1.0, atol=1e-5)
    assert np.array_equal(x, oracle.synth_rows(4000, 96))            # deterministic
    assert np.array_equal(x[1000:1500], oracle.synth_rows(4000, 96, 1000, 500))  # addressable by row
    dup_rows = [r for r in range(4000) if r % 1000 == 999]
    for r in dup_rows:
        assert (x == x[r]).all(axis=1).sum() >= 2                     # exact duplicates exist
    near = x[998] @ x.T
    assert np.sort(near)[-2] > 0.995                                  # near-duplicate partner
    xs = oracle.synth_rows(2000, 96, flags=3)
    n = np.linalg.norm(xs, axis=1)
    assert n.min() >= 0.49 and n.max() <= 2.01 and n.std() > 0.2     # un-normalised fixture


def test_dedup_scan_and_autolink_semantics(oracle):
    # rows 0,1,2 mutually similar (>0.92); row 3 unrelated; row 1 tombstoned in storage (Q2)
    ix = oracle.OracleIndex(3)
    v = np.array([[1, 0, 0], [0.99, 0.05, 0], [0.98, 0, 0.06], [0, 1, 0]], np.float32)
    ix.insert_batch(ids_for(4), v)
    pairs = ix.dedup_scan(0.92)
    assert sorted((int(p["from_row"]), int(p["to_row"])) for p in pairs) == [(0, 1), (0, 2), (1, 2)]
    deleted = np.array([0, 1, 0, 0], np.uint8)
    pairs = ix.dedup_scan(0.92, deleted)
    # deleted nodes are not scanned but still appear as neighbours (dedup.rs:72-74, :105-108)
    assert sorted((int(p["from_row"]), int(p["to_row"])) for p in pairs) == [(0, 1), (0, 2), (2, 1)]
    edges = ix.autolink_pass([0, 3], 100, 0.75, 1, deleted)
    # per-node cap 1 (auto_linker.rs:261-263); deleted neighbour skipped (:240-243)
    assert [(int(e["from_row"]), int(e["to_row"])) for e in edges] == [(0, 2)]


def test_hnsw_baseline_restatement_has_high_recall(oracle):
    """The CPU HNSW baseline is only a reported number (parity unpinned); this checks it is a working ANN."""
    n, d, k = 3000, 64, 10
    rows = oracle.synth_rows(n, d)
    qs = oracle.synth_queries(n, d, 20)
    ix = oracle.OracleIndex(d)
    ix.insert_batch(ids_for(n), rows)
    h = oracle.HnswBaseline(rows)
    hits = 0
    for q in qs:
        r, dist = h.search(q, k)
        e = ix.search(q, k)
        hits += len(set(r.tolist()) & set(e["row"].tolist()))
        assert np.all(np.diff(dist) >= 0)
    assert hits / (len(qs) * k) > 0.9
    # the concurrent build + parallel batch search (what bench.py times on all host cores): a working ANN too
    hm = oracle.HnswBaseline(rows, n_threads=4)
    r, dist, cnt = hm.search_batch(qs, k, 100, n_threads=4)
    hits = 0
    for i, q in enumerate(qs):
        e = ix.search(q, k)
        hits += len(set(r[i, :cnt[i]].tolist()) & set(e["row"].tolist()))
        assert np.all(np.diff(dist[i, :cnt[i]]) >= 0)
    assert hits / (len(qs) * k) > 0.9
