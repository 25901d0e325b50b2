"""bf16 row stores (cx_create_ex, CX_DTYPE_BF16; BASELINE config 5's storage dtype): every vector is rounded to bf16 once,
at insert, and every result must be what the reference's exact path returns for the ROUNDED vectors — so the oracle is
fed bf16_round(rows) as f32 and everything is compared as in the f32 tests (ids exact up to near-ties, scores within
SCORE_TOL).  Covers the paths that read the store: single scan (fixed-dim and generic kernels), threshold search,
batched search (batchg.hip's bf16-row instance, dense and bound + candidates output), the all-pairs passes (shadow from
bf16 rows, exact rescore on bf16 rows), upsert / remove / rebuild, save / load, and the sharded index."""
import os

import numpy as np
import pytest

from conftest import SCORE_TOL, assert_topk_parity, ids_for

pytestmark = pytest.mark.gpu


def bf16_round(x: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).reshape(x.shape)


def rows_of(index_ids, got_ids):
    lut = {index_ids[i].tobytes(): i for i in range(len(index_ids))}
    return np.array([lut[g.tobytes()] for g in got_ids], dtype=np.int64)


def build_both(hip, oracle, rows):
    ids = ids_for(len(rows))
    h = hip.HipIndex(rows.shape[1], dtype="bf16")
    h.insert_batch(ids, rows)                 # the engine rounds
    o = oracle.OracleIndex(rows.shape[1])
    o.insert_batch(ids, bf16_round(rows))     # the oracle is given the rounded rows
    return h, o, ids


@pytest.mark.parametrize("n,d,k", [(3000, 768, 10), (2500, 384, 5), (3000, 1024, 10), (700, 100, 7), (900, 128, 100),
                                   (1200, 1536, 3), (400, 256, 300), (1000, 512, 1), (50, 3, 5)])
def test_single_search_matches_oracle_on_rounded_rows(hip, oracle, n, d, k):
    rows = oracle.synth_rows(n, d)
    qs = oracle.synth_queries(n, d, 12)
    h, o, ids = build_both(hip, oracle, rows)
    assert len(h) == n
    for i, q in enumerate(qs):
        gi, gs, gd = h.search_arrays(q, k)
        e = o.search(q, k)
        assert len(gs) == len(e["row"])
        assert_topk_parity(rows_of(ids, gi), gs, e["row"], e["score"], what=f"bf16 n={n} d={d} k={k} q{i}")
        assert np.all(np.abs(gd - (1.0 - gs)) <= 2e-6) or np.any(gs == 0.0)
    # the rounding is real: an f32 store scores differently somewhere
    f = hip.HipIndex(d)
    f.insert_batch(ids, rows)
    _, fs, _ = f.search_arrays(qs[0], min(k, 5))
    _, bs, _ = h.search_arrays(qs[0], min(k, 5))
    if d >= 100:
        assert np.any(fs != bs)


def test_threshold_filters_removes_and_rebuild(hip, oracle):
    n, d = 4000, 768
    rows = oracle.synth_rows(n, d)
    qs = oracle.synth_queries(n, d, 8)
    h, o, ids = build_both(hip, oracle, rows)
    for r in range(0, n, 3):
        kind = "fact" if r % 2 else "event"
        h.set_metadata(ids[r].tobytes(), kind, "kai"); o.set_metadata(ids[r].tobytes(), kind, "kai")
    for r in (10, 11, 500, 3999):
        h.remove(ids[r].tobytes()); o.remove(ids[r].tobytes())
    excl = [ids[i].tobytes() for i in (1, 2, 3)]
    hf, of = hip.VectorFilter(kinds=["fact"], exclude=excl), oracle.Filter(kinds=["fact"], exclude=excl)
    for q in qs:
        gi, gs, gd = h.search_threshold_arrays(q, 0.6)
        e = o.search_threshold(q, 0.6)
        assert len(gs) == len(e["row"])
        assert_topk_parity(rows_of(ids, gi), gs, e["row"], e["score"], what="bf16 threshold")
        gi, gs, gd = h.search_arrays(q, 10, hf)
        e = o.search(q, 10, of)
        assert_topk_parity(rows_of(ids, gi), gs, e["row"], e["score"], what="bf16 filtered")
    # upsert in place (a known id keeps its row) and a rebuild that compacts the bf16 rows
    new = rows[100:105][::-1].copy() * np.float32(1.7)
    for t, r in enumerate((20, 21, 22, 23, 24)):
        h.insert(ids[r].tobytes(), new[t]); o.insert(ids[r].tobytes(), bf16_round(new[t]))
    h.rebuild(); o.rebuild()
    assert len(h) == len(o) == n - 4
    for q in qs[:4]:
        gi, gs, gd = h.search_arrays(q, 10)
        e = o.search(q, 10)
        assert len(gs) == len(e["score"]) and np.allclose(gs, e["score"], atol=SCORE_TOL)


@pytest.mark.parametrize("n,d,k,nq", [(5000, 1024, 10, 64), (3001, 768, 100, 70), (777, 512, 32, 7), (40, 1024, 40, 70),
                                      (1300, 640, 10, 9), (2100, 384, 10, 33), (900, 2048, 10, 70), (300, 4096, 5, 9), (257, 128, 256, 64)])
def test_search_batch_matches_oracle_on_rounded_rows(hip, oracle, n, d, k, nq):
    rows = oracle.synth_rows(n, d)
    qs = oracle.synth_queries(n, d, nq)
    h, o, ids = build_both(hip, oracle, rows)
    bi, bs, bd, bc = h.search_batch_arrays(qs, k)
    for i in range(nq):
        e = o.search(qs[i], k)
        m = int(bc[i])
        assert m == len(e["row"])
        assert_topk_parity(rows_of(ids, bi[i, :m]), bs[i, :m], e["row"], e["score"], what=f"bf16 batch n={n} d={d} k={k} q{i}")


def test_batched_search_at_scale_matches_single_scans(hip):
    """300k x 768 bf16 (0.46 GB): the bound + candidates output of batchg.hip's bf16-row instance against single-query
    scans of the same store (pinned to the oracle above)."""
    import torch
    from cortex_amd import _lib
    L = _lib.load()
    n, d, nq = 300_000, 768, 70
    gen = torch.empty((n, d), dtype=torch.float32, device="cuda:0")
    assert L.cx_synth_fill_dev(0, gen.data_ptr(), 20260313, 20260313, 20260315, n // 50, 0, n, d, 1) == 0
    ids = np.zeros((n, 16), np.uint8)
    ids[:, 8:] = np.arange(n, dtype=np.uint64).astype(">u8").view(np.uint8).reshape(n, 8)
    h = hip.HipIndex(d, dtype="bf16")
    h.insert_batch_dev(ids, gen.data_ptr(), n, d)
    qs_t = torch.empty((nq, d), dtype=torch.float32, device="cuda:0")
    assert L.cx_synth_fill_dev(0, qs_t.data_ptr(), 20260313, 20260314, 20260315, n // 50, 0, nq, d, 0) == 0
    qs = qs_t.cpu().numpy()
    del gen
    for k in (10, 100):
        bi, bs, bd, bc = h.search_batch_arrays(qs, k)
        for i in range(nq):
            gi, gs, gd = h.search_arrays(qs[i], k)
            m = int(bc[i])
            assert m == len(gs) == k
            got = np.array([int.from_bytes(bytes(x[8:]), "big") for x in bi[i, :m]])
            exp = np.array([int.from_bytes(bytes(x[8:]), "big") for x in gi])
            assert_topk_parity(got, bs[i, :m], exp, gs, what=f"bf16 k={k} q{i}")


@pytest.mark.parametrize("n,d", [(2000, 768), (1200, 1024), (700, 100)])
def test_autolink_pass_on_rounded_rows(hip, oracle, n, d):
    from test_hip_autolink import compare_edges, oracle_scores, per_node   # the comparison of the f32 passes
    rows = oracle.synth_rows(n, d)
    h, o, ids = build_both(hip, oracle, rows)
    thr = float(np.float32(0.85))
    fr, to, w = h.autolink_pass_rows(None, 100, thr, 50)
    e = o.autolink_pass(np.arange(n), 100, thr, 50, n_threads=8)
    got, exp = per_node(fr, to, w), per_node(e["from_row"], e["to_row"], e["weight"])
    assert len(exp) > 0
    compare_edges(got, exp, thr, oracle_scores(o, bf16_round(rows)), f"bf16 n={n} d={d}")
    a, b, s = h.dedup_scan_rows(float(np.float32(0.92)))
    ed = o.dedup_scan(float(np.float32(0.92)))
    assert abs(len(a) - len(ed)) <= max(2, len(ed) // 200)   # pairs within SCORE_TOL of the threshold may differ


@pytest.mark.parametrize("batch", [64, 500])
def test_streaming_ingest_tick_links_the_appended_rows(hip, oracle, batch):
    """BASELINE configs[4] as the reference runs it (auto_linker.rs:378-398 -> :220-221): a batch of NEW rows is inserted
    into a bf16 store (cx_upsert_batch_dev: rounded once, on the device), the shadow and its tiled copy grow by exactly
    those rows, and the pass links exactly those rows against the whole store — their own batch included.  Edges must be
    the reference's for the rounded vectors.  64 rows take the streaming filter kernel, 500 the 256-tile kernel; two
    ticks in a row, with an in-place upsert of an old row between them (the shadow's stale list)."""
    import torch
    from test_hip_autolink import compare_edges, oracle_scores, per_node
    n0, d = 3000, 1024
    allrows = oracle.synth_rows(n0 + 2 * batch, d)
    ids = ids_for(n0 + 2 * batch)
    h = hip.HipIndex(d, dtype="bf16")
    h.reserve(n0 + 2 * batch)
    h.insert_batch(ids[:n0], allrows[:n0])
    o = oracle.OracleIndex(d)
    o.insert_batch(ids[:n0], bf16_round(allrows[:n0]))
    thr = float(np.float32(0.85))
    h.autolink_pass_rows(np.arange(n0 - 64, n0, dtype=np.uint32), 100, thr, 50)       # the shadow exists before the ticks
    lo = n0
    for tick in range(2):
        new = allrows[lo:lo + batch]
        dev = torch.from_numpy(new).to("cuda:0")
        h.insert_batch_dev(ids[lo:lo + batch], dev.data_ptr(), batch, d)
        o.insert_batch(ids[lo:lo + batch], bf16_round(new))
        scan = np.arange(lo, lo + batch, dtype=np.uint32)
        fr, to, w = h.autolink_pass_rows(scan, 100, thr, 50)
        e = o.autolink_pass(scan, 100, thr, 50, n_threads=8)
        got, exp = per_node(fr, to, w), per_node(e["from_row"], e["to_row"], e["weight"])
        assert len(exp) > 0 and set(got) <= set(int(x) for x in scan)
        compare_edges(got, exp, thr, oracle_scores(o, bf16_round(allrows[:lo + batch])), f"ingest tick {tick} batch {batch}")
        assert any(int(t) >= lo for t in to), "no edge inside the new batch: the pass did not see the appended rows as neighbours"
        lo += batch
        if tick == 0:     # an old row changes in place: its shadow row (and tiled copy) must be refreshed before the next tick
            allrows[7] = allrows[lo + 3] * np.float32(1.5)      # now a near-copy of a row of the NEXT batch
            h.insert(ids[7].tobytes(), allrows[7])
            o.insert(ids[7].tobytes(), bf16_round(allrows[7:8])[0])
    assert 7 in set(int(t) for t in to), "the upserted row is not a neighbour of its near-copy: stale shadow"


def test_save_load_round_trip(hip, oracle, tmp_path):
    n, d = 1500, 384
    rows = oracle.synth_rows(n, d)
    qs = oracle.synth_queries(n, d, 4)
    h, o, ids = build_both(hip, oracle, rows)
    p = str(tmp_path / "bf16.idx")
    h.save(p)
    f = hip.HipIndex.load(p)                       # the file holds the rounded values as f32
    b = hip.HipIndex.load(p, dtype="bf16")         # rounding them again changes nothing
    for q in qs:
        _, s0, _ = h.search_arrays(q, 10)
        _, s1, _ = f.search_arrays(q, 10)
        _, s2, _ = b.search_arrays(q, 10)
        assert np.allclose(s0, s1, atol=SCORE_TOL) and np.array_equal(s0, s2)


def test_sharded_bf16_equals_single_bf16(hip, oracle):
    n, d = 5000, 768
    rows = oracle.synth_rows(n, d)
    qs = oracle.synth_queries(n, d, 40)
    ids = ids_for(n)
    one = hip.HipIndex(d, dtype="bf16"); one.insert_batch(ids, rows)
    sh = hip.ShardedHipIndex(d, [0, 0, 0], dtype="bf16"); sh.insert_batch(ids, rows)
    for k in (10, 100):
        ai, asc, ad, ac = one.search_batch_arrays(qs, k)
        bi, bsc, bd, bc = sh.search_batch_arrays(qs, k)
        assert np.array_equal(ac, bc)
        for i in range(len(qs)):
            assert_topk_parity(rows_of(ids, bi[i, :int(bc[i])]), bsc[i, :int(bc[i])], rows_of(ids, ai[i, :int(ac[i])]), asc[i, :int(ac[i])], what=f"sharded bf16 k={k} q{i}")
    f1, t1, w1 = one.autolink_pass_rows(None, 100, 0.85, 50)
    f2, t2, w2 = sh.autolink_pass_rows(None, 100, 0.85, 50)
    assert len(f1) == len(f2) and np.array_equal(f1, f2) and np.array_equal(t1, t2) and np.allclose(w1, w2, atol=SCORE_TOL)
