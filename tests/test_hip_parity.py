"""Parity of the HIP path (through the C ABI) with the CPU oracle and the golden fixtures.

Bar (north_star): row ids bit-exact under the declared order (score desc, row asc, NaN last)
except among near-ties closer than SCORE_TOL; scores/distances within SCORE_TOL = 5e-5 absolute
(conftest.py explains the bound)."""
import glob
import os
import uuid

import numpy as np
import pytest

from conftest import SCORE_TOL, assert_topk_parity, ids_for

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def rows_of(index_ids: np.ndarray, got_ids: np.ndarray) -> np.ndarray:
    """map returned 16-byte ids back to insertion rows"""
    lut = {index_ids[i].tobytes(): i for i in range(len(index_ids))}
    return np.array([lut[g.tobytes()] for g in got_ids], dtype=np.int64)


def build_both(hip, oracle, rows, ids=None):
    ids = ids_for(len(rows)) if ids is None else ids
    h = hip.HipIndex(rows.shape[1])
    h.insert_batch(ids, rows)
    o = oracle.OracleIndex(rows.shape[1])
    o.insert_batch(ids, rows)
    return h, o, ids


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_golden_fixtures(hip, path):
    g = np.load(path)
    rows, qs, k = g["rows"], g["queries"], int(g["k"])
    ids = ids_for(len(rows))
    h = hip.HipIndex(rows.shape[1])
    h.insert_batch(ids, rows)
    for i, q in enumerate(qs):
        gi, gs, gd = h.search_arrays(q, k)
        n = len(gs)
        assert_topk_parity(rows_of(ids, gi), gs, g["exp_rows"][i][:n], g["exp_scores"][i][:n], what=f"{path} q{i}")
        exp_d = g["exp_dists"][i][:n]
        pos = {int(r): j for j, r in enumerate(g["exp_rows"][i][:n])}
        for j, r in enumerate(rows_of(ids, gi)):
            if int(r) in pos:
                e = exp_d[pos[int(r)]]
                assert (np.isnan(e) and np.isnan(gd[j])) or abs(e - gd[j]) <= SCORE_TOL
    # the batch entry point (MFMA path for 384/768-d, k <= 32) meets the same bar; its summation order
    # differs from the single-query kernel's, so scores may differ from it in the last ulp
    bi, bs, bd, bc = h.search_batch_arrays(qs, k)
    for i in range(len(qs)):
        n = int(bc[i])
        assert_topk_parity(rows_of(ids, bi[i, :n]), bs[i, :n], g["exp_rows"][i][:n], g["exp_scores"][i][:n],
                           what=f"{path} batch q{i}")


@pytest.mark.parametrize("n,d,k", [
    (1000, 384, 5),      # BASELINE config 1 shape, reduced rows
    (5000, 768, 10),     # headline dim
    (3000, 1024, 10),    # config 5 dim
    (2000, 128, 100),    # k = 100 (auto-linker's k), G=32 kernel
    (777, 256, 200),     # k in the 4-slot lists, ragged tail
    (513, 100, 10),      # dim with no fast kernel -> generic path
    (300, 7, 3),         # tiny odd dim
    (64, 1536, 64),      # k == n
])
def test_search_matches_oracle(hip, oracle, n, d, k):
    rows = oracle.synth_rows(n, d)
    qs = oracle.synth_queries(n, d, 6)
    h, o, ids = build_both(hip, oracle, rows)
    for i, q in enumerate(qs):
        gi, gs, gd = h.search_arrays(q, k)
        e = o.search(q, k)
        assert_topk_parity(rows_of(ids, gi), gs, e["row"], e["score"], what=f"n={n} d={d} k={k} q{i}")
        assert np.all(np.abs(gd - (1.0 - (1.0 - gd))) < 1e-6)  # distance finite and consistent
        assert np.all(np.diff(gs) <= 0), "scores not sorted descending"


def test_unnormalised_rows_and_queries(hip, oracle):
    rows = oracle.synth_rows(1500, 384, flags=3)  # norms in [0.5, 2)
    qs = oracle.synth_queries(1500, 384, 4) * np.float32(3.25)
    h, o, ids = build_both(hip, oracle, rows)
    for q in qs:
        gi, gs, gd = h.search_arrays(q, 10)
        e = o.search(q, 10)
        assert_topk_parity(rows_of(ids, gi), gs, e["row"], e["score"], what="unnormalised")


def test_exact_duplicates_resolve_by_insertion_row(hip, oracle):
    rows = oracle.synth_rows(3000, 256)            # rows 999, 1999, 2999 duplicate earlier rows
    h, o, ids = build_both(hip, oracle, rows)
    for r in (999, 1999, 2999):
        gi, gs, gd = h.search_arrays(rows[r], 4)
        e = o.search(rows[r], 4)
        got = rows_of(ids, gi)
        assert list(got[:2]) == list(e["row"][:2]), "duplicate pair must come back in insertion order"
        assert got[0] < got[1] and gs[0] == gs[1]


def test_large_k_and_threshold_paths(hip, oracle):
    rows = oracle.synth_rows(1200, 384)
    q = oracle.synth_queries(1200, 384, 1)[0]
    h, o, ids = build_both(hip, oracle, rows)
    # k > 256 -> sort path; k > len -> everything
    for k in (300, 1200, 5000):
        gi, gs, gd = h.search_arrays(q, k)
        e = o.search(q, k)
        assert_topk_parity(rows_of(ids, gi), gs, e["row"], e["score"], what=f"k={k}")
    for thr in (0.92, 0.75, 0.3, 0.0, -1.0, 1.5):
        gi, gs, gd = h.search_threshold_arrays(q, thr)
        e = o.search_threshold(q, thr)
        # rows whose oracle score is within tol of thr may fall either side
        edge = np.abs(o.search(q, 1200)["score"].astype(np.float64) - thr) <= SCORE_TOL
        assert abs(len(gs) - len(e)) <= edge.sum(), f"thr={thr}: {len(gs)} vs {len(e)}"
        if len(gs) == len(e):
            assert_topk_parity(rows_of(ids, gi), gs, e["row"], e["score"], what=f"thr={thr}")
        assert np.all(gs >= np.float32(thr))


def test_filters_match_oracle(hip, oracle):
    n, d = 900, 128
    rows = oracle.synth_rows(n, d)
    q = oracle.synth_queries(n, d, 1)[0]
    h, o, ids = build_both(hip, oracle, rows)
    kinds = ["fact", "decision", "event"]
    agents = ["kai", "test"]
    for r in range(0, n, 2):           # only even rows get metadata: "no metadata => passes" (Q3)
        k, a = kinds[r % 3], agents[(r // 2) % 2]
        h.set_metadata(ids[r].tobytes(), k, a)
        o.set_metadata(ids[r].tobytes(), k, a)
    top = o.search(q, 5)
    excl = [bytes(x["node_id"]) for x in top[:3]] + [uuid.uuid4().bytes]  # plus an id that is not indexed
    cases = [
        (hip.VectorFilter(kinds=["decision"]), oracle.Filter(kinds=["decision"])),
        (hip.VectorFilter(kinds=[]), oracle.Filter(kinds=[])),
        (hip.VectorFilter(kinds=["fact", "event"], source_agent="kai"), oracle.Filter(kinds=["fact", "event"], source_agent="kai")),
        (hip.VectorFilter(exclude=excl), oracle.Filter(exclude=excl)),
        (hip.VectorFilter(exclude=excl, kinds=["event"], source_agent="nobody"), oracle.Filter(exclude=excl, kinds=["event"], source_agent="nobody")),
    ]
    for hf, of in cases:
        for k in (7, 600):
            gi, gs, gd = h.search_arrays(q, k, hf)
            e = o.search(q, k, of)
            assert_topk_parity(rows_of(ids, gi), gs, e["row"], e["score"], what=f"filter {hf} k={k}")
        gi, gs, gd = h.search_threshold_arrays(q, 0.5, hf)
        e = o.search_threshold(q, 0.5, of)
        assert abs(len(gs) - len(e)) <= 1


def test_upsert_remove_rebuild_semantics(hip, oracle):
    n, d = 400, 64
    rows = oracle.synth_rows(n, d)
    q = rows[10]
    h, o, ids = build_both(hip, oracle, rows)
    # upsert replaces in place, visible immediately
    new = oracle.synth_queries(n, d, 1)[0]
    h.insert(ids[10].tobytes(), new)
    o.insert(ids[10].tobytes(), new)
    assert len(h) == len(o) == n
    for k in (5, n):
        gi, gs, _ = h.search_arrays(q, k)
        e = o.search(q, k)
        assert_topk_parity(rows_of(ids, gi), gs, e["row"], e["score"], what="after upsert")
    # remove: gone from results at once; len drops
    victims = [int(x) for x in o.search(q, 3)["row"]]
    for v in victims:
        h.remove(ids[v].tobytes())
        o.remove(ids[v].tobytes())
    h.remove(uuid.uuid4().bytes)  # unknown id: not an error (index.rs:316-323)
    assert len(h) == len(o) == n - 3
    gi, gs, _ = h.search_arrays(q, 10)
    e = o.search(q, 10)
    assert not set(victims) & set(rows_of(ids, gi))
    assert_topk_parity(rows_of(ids, gi), gs, e["row"], e["score"], what="after remove")
    # rebuild compacts; results unchanged; re-insert of a removed id lands at the end
    h.rebuild()
    assert h.row_count() == n - 3
    gi2, gs2, _ = h.search_arrays(q, 10)
    assert np.array_equal(gi, gi2) and np.array_equal(gs, gs2)
    h.insert(ids[victims[0]].tobytes(), rows[victims[0]])
    o.insert(ids[victims[0]].tobytes(), rows[victims[0]])
    gi, gs, _ = h.search_arrays(rows[victims[0]], 2)
    assert gi[0].tobytes() == ids[victims[0]].tobytes() and gs[0] > 0.9999
    assert h.row_id(h.row_count() - 1).bytes == ids[victims[0]].tobytes()


def test_query_length_mismatch_is_not_an_error(hip, oracle):
    """The reference zips (index.rs:172): a longer/shorter query still searches."""
    from oracle import np_twin as T
    rows = oracle.synth_rows(200, 64)
    ids = ids_for(200)
    h = hip.HipIndex(64)
    h.insert_batch(ids, rows)
    rng = np.random.default_rng(5)
    for qlen in (40, 100):
        q = rng.standard_normal(qlen).astype(np.float32)
        gi, gs, gd = h.search_arrays(q, 5)
        m = min(qlen, 64)
        dot = np.array([np.cumsum(r[:m] * q[:m], dtype=np.float32)[-1] for r in rows])
        nr = np.sqrt(np.array([np.cumsum(r * r, dtype=np.float32)[-1] for r in rows]))
        nq = np.sqrt(np.cumsum(q * q, dtype=np.float32)[-1])
        s = np.clip(1 - (1 - dot / (nq * nr)), 0, 1)
        order = np.argsort(-s, kind="stable")[:5]
        assert_topk_parity(rows_of(ids, gi), gs, order, s[order], what=f"qlen={qlen}")


def test_zero_norm_rows_and_query_give_nan_last(hip, oracle):
    rows = oracle.synth_rows(50, 32).copy()
    rows[7] = 0.0
    h, o, ids = build_both(hip, oracle, rows)
    gi, gs, gd = h.search_arrays(rows[3], 50)
    e = o.search(rows[3], 50)
    assert np.isnan(gs[-1]) and rows_of(ids, gi)[-1] == 7
    assert_topk_parity(rows_of(ids, gi), gs, e["row"], e["score"], what="zero row")
    gi, gs, gd = h.search_arrays(np.zeros(32, np.float32), 5)   # zero query: all NaN, row order
    assert np.all(np.isnan(gs)) and list(rows_of(ids, gi)) == [0, 1, 2, 3, 4]
    assert len(h.search_threshold_arrays(np.zeros(32, np.float32), 0.0)[1]) == 0  # NaN >= t is false


def test_incremental_inserts_visible_without_rebuild(hip, oracle):
    d = 384
    rows = oracle.synth_rows(600, d)
    ids = ids_for(600)
    h = hip.HipIndex(d)
    o = oracle.OracleIndex(d)
    q = oracle.synth_queries(600, d, 1)[0]
    for lo in range(0, 600, 150):   # four batches, a search after each, never a rebuild (Q1)
        h.insert_batch(ids[lo:lo + 150], rows[lo:lo + 150])
        o.insert_batch(ids[lo:lo + 150], rows[lo:lo + 150])
        gi, gs, _ = h.search_arrays(q, 10)
        e = o.search(q, 10)
        assert_topk_parity(rows_of(ids, gi), gs, e["row"], e["score"], what=f"after {lo + 150} rows")


def test_concurrent_readers(hip, oracle):
    """&self methods are re-entrant: many threads search at once (RwLock read side)."""
    import threading
    rows = oracle.synth_rows(4000, 384)
    qs = oracle.synth_queries(4000, 384, 16)
    h, o, ids = build_both(hip, oracle, rows)
    want = [o.search(q, 10) for q in qs]
    errs = []

    def work(t):
        try:
            for rep in range(5):
                for i in range(t, len(qs), 4):
                    gi, gs, _ = h.search_arrays(qs[i], 10)
                    assert_topk_parity(rows_of(ids, gi), gs, want[i]["row"], want[i]["score"], what=f"thread {t}")
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs[0]


def test_concurrent_filtered_readers(hip, oracle):
    """Filters on the read side: kinds and source_agent strings are looked up (cx_lookup, read-only), never interned,
    so concurrent filtered searches — one of them carrying strings nobody was tagged with — do not touch the table."""
    import threading
    n, d = 3000, 384
    rows = oracle.synth_rows(n, d)
    qs = oracle.synth_queries(n, d, 12)
    h, o, ids = build_both(hip, oracle, rows)
    kinds, agents = ["fact", "decision", "event"], ["kai", "test"]
    for r in range(0, n, 2):
        h.set_metadata(ids[r].tobytes(), kinds[r % 3], agents[(r // 2) % 2])
        o.set_metadata(ids[r].tobytes(), kinds[r % 3], agents[(r // 2) % 2])
    cases = [(hip.VectorFilter(kinds=["decision"]), oracle.Filter(kinds=["decision"])),
             (hip.VectorFilter(kinds=["fact", "event"], source_agent="kai"), oracle.Filter(kinds=["fact", "event"], source_agent="kai"))]
    want = [[o.search(q, 10, of) for q in qs] for _, of in cases]
    errs = []

    def work(t):
        try:
            for rep in range(6):
                for i in range(len(qs)):
                    if t == 3:   # fresh strings every call: with cx_intern on this path the table would grow under the others
                        f = hip.VectorFilter(kinds=[f"kind-{rep}-{i}"], source_agent=f"agent-{rep}-{i}")
                        gi, gs, _ = h.search_arrays(qs[i], 10, f)
                        assert all(rows_of(ids, gi) % 2 == 1)            # only rows without metadata pass
                    else:
                        hf, _ = cases[t % 2]
                        gi, gs, _ = h.search_arrays(qs[i], 10, hf)
                        e = want[t % 2][i]
                        assert_topk_parity(rows_of(ids, gi), gs, e["row"], e["score"], what=f"thread {t}")
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    before = h.lookup("kind-0-0")
    ts = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs[0]
    assert before == 0 and h.lookup("kind-0-0") == 0 and h.lookup("agent-5-11") == 0


@pytest.mark.parametrize("n,d,k,nq", [
    (5000, 768, 10, 64),     # BASELINE config 4's inner loop: batch-64, k=10
    (3001, 384, 5, 100),     # two passes (64 + 36), ragged last tile
    (777, 768, 32, 7),       # k = 32 (largest fused k), fewer queries than one wave's share
    (40, 384, 10, 3),        # fewer rows than blocks
    (2000, 768, 100, 5),     # k > 32: falls back to one scan per query
    (40, 384, 40, 70),       # wide lists for all 64 queries over fewer rows than one block's first tiles (k == n)
    (300, 768, 100, 64),     # the same at 768-d (single tile buffer), one full group
    (17, 768, 17, 33),       # a corpus of one ragged tile
    (5000, 1024, 10, 64),    # 1024-d (BGE-large, config 5's width): batchg.hip — queries through LDS, rows split in registers
    (3001, 1024, 100, 70),   # two passes (64 + 6), ragged last row tile, the linker's k
    (777, 512, 32, 7),
    (1300, 1536, 10, 3),     # three queries: the smallest batch that takes the batched path
    (130, 256, 5, 40),       # one K-block pair per row; fewer rows than one row tile
    (2000, 256, 256, 5),     # the largest in-register k
    (40, 1024, 40, 70),      # k == n: fewer rows than one tile
    (1300, 640, 10, 9),      # dim % 128 == 0 is all batchg.hip asks for
    (700, 128, 64, 33),
    (900, 2048, 10, 70),     # the widest rows batchg.hip takes
    (300, 4096, 5, 9),
])
def test_search_batch_matches_oracle(hip, oracle, n, d, k, nq):
    rows = oracle.synth_rows(n, d)
    qs = oracle.synth_queries(n, d, nq)
    h, o, ids = build_both(hip, oracle, rows)
    bi, bs, bd, bc = h.search_batch_arrays(qs, k)
    for i in range(nq):
        e = o.search(qs[i], k)
        m = int(bc[i])
        assert_topk_parity(rows_of(ids, bi[i, :m]), bs[i, :m], e["row"], e["score"], what=f"batch n={n} d={d} k={k} q{i}")
        assert np.all(np.diff(bs[i, :m]) <= 0)
        pos = {int(r): j for j, r in enumerate(e["row"])}
        for jj, r in enumerate(rows_of(ids, bi[i, :m])):
            if int(r) in pos:
                assert abs(bd[i, jj] - e["distance"][pos[int(r)]]) <= SCORE_TOL


def test_search_batch_with_filter_and_tombstones(hip, oracle):
    n, d = 1500, 768
    rows = oracle.synth_rows(n, d)
    qs = oracle.synth_queries(n, d, 20)
    h, o, ids = build_both(hip, oracle, rows)
    for r in range(0, n, 3):
        h.set_metadata(ids[r].tobytes(), "fact" if r % 2 else "event", "kai")
        o.set_metadata(ids[r].tobytes(), "fact" if r % 2 else "event", "kai")
    for r in (10, 11, 500):
        h.remove(ids[r].tobytes()); o.remove(ids[r].tobytes())
    excl = [ids[i].tobytes() for i in (1, 2, 3)]
    hf, of = hip.VectorFilter(kinds=["fact"], exclude=excl), oracle.Filter(kinds=["fact"], exclude=excl)
    bi, bs, bd, bc = h.search_batch_arrays(qs, 10, hf)
    for i in range(len(qs)):
        e = o.search(qs[i], 10, of)
        m = int(bc[i])
        assert_topk_parity(rows_of(ids, bi[i, :m]), bs[i, :m], e["row"], e["score"], what=f"batch filter q{i}")


@pytest.mark.parametrize("n,d,k,nq", [
    (60000, 384, 10, 64),    # ~15 tiles per block: lists fill, producers compact, blocks share their bound
    (50000, 768, 32, 20),    # k = 32: compaction threshold == k + 16, every compaction keeps 2/3 of the list
    (33000, 768, 1, 33),     # k = 1, ragged query group
    (50000, 768, 100, 40),   # wide mode: the auto-linker's top-100 lists, 32 queries per pass, 3 entries per lane
    (20000, 384, 104, 33),   # largest fused k; second pass holds one query
    (9000, 768, 33, 5),      # smallest wide k
    (30000, 384, 10, 200),   # several query groups in one launch (grid = chunks x groups), ragged last group
    (30000, 768, 100, 70),   # the same in the wide mode: 3 groups of 32, the last holds 6
    (60000, 384, 100, 150),  # wide lists at 384-d keep all 64 queries per pass: 3 groups, the last holds 22
    (25000, 384, 64, 64),    # one full 64-query wide group
])
def test_search_batch_long_lists(hip, oracle, n, d, k, nq):
    """Enough rows per block that the in-kernel candidate lists overflow and are compacted many times
    (batch.hip: producer_compact / apply_shrink / global slots)."""
    rows = oracle.synth_rows(n, d)
    qs = oracle.synth_queries(n, d, nq)
    h, o, ids = build_both(hip, oracle, rows)
    for r in (5, 17, 40000 % n):
        h.remove(ids[r].tobytes()); o.remove(ids[r].tobytes())
    bi, bs, bd, bc = h.search_batch_arrays(qs, k)
    for i in range(nq):
        e = o.search(qs[i], k)
        m = int(bc[i])
        assert m == len(e["row"])
        assert_topk_parity(rows_of(ids, bi[i, :m]), bs[i, :m], e["row"], e["score"], what=f"batch long n={n} k={k} q{i}")


@pytest.mark.parametrize("k", [32, 100])
def test_search_batch_massive_ties(hip, oracle, k):
    """50 distinct vectors repeated 800 times: almost every cut falls inside a run of equal scores, so the
    compaction's tie rule (lower insertion row wins) decides the result.  ids must match exactly."""
    n, d = 40000, 384
    base = oracle.synth_rows(50, d)
    rows = np.ascontiguousarray(base[np.arange(n) % 50])
    qs = np.ascontiguousarray(base[:16] + 0.05 * oracle.synth_queries(50, d, 16))
    h, o, ids = build_both(hip, oracle, rows)
    bi, bs, bd, bc = h.search_batch_arrays(qs, k)
    for i in range(len(qs)):
        e = o.search(qs[i], k)
        m = int(bc[i])
        assert m == k
        got = rows_of(ids, bi[i, :m])
        # equal vectors give bit-equal scores on both sides, so the order is fully determined
        assert list(map(int, got)) == list(map(int, e["row"])), f"q{i}: {got[:8]} vs {e['row'][:8]}"


def test_search_batch_after_upserts_removes_and_rebuild(hip, oracle):
    """The batched search reads |row|^2 from a cache kept next to the rows: in-place upserts, appends, removes
    and rebuild (compaction) must all leave it in step with the store."""
    n, d, k = 6000, 768, 10
    rows = oracle.synth_rows(n + 500, d)
    qs = oracle.synth_queries(n, d, 70)
    ids = ids_for(n + 500)
    h = hip.HipIndex(d); o = oracle.OracleIndex(d)
    h.insert_batch(ids[:n], rows[:n]); o.insert_batch(ids[:n], rows[:n])

    def check(tag):
        bi, bs, bd, bc = h.search_batch_arrays(qs, k)
        for i in range(len(qs)):
            e = o.search(qs[i], k)
            m = int(bc[i])
            got_ids = [bytes(x) for x in bi[i, :m]]
            want_ids = [bytes(x) for x in e["node_id"]]
            assert m == len(want_ids), tag
            # ids exact except near-ties
            for j, (g, w) in enumerate(zip(got_ids, want_ids)):
                if g != w:
                    assert abs(float(bs[i, j]) - float(e["score"][j])) <= SCORE_TOL, f"{tag} q{i} pos {j}"
            assert np.allclose(bs[i, :m], e["score"], atol=SCORE_TOL), tag

    check("fresh")
    # in-place upserts: scaled copies of other rows (norms change by 4x and 0.25x)
    for r, src, sc in ((5, 100, 2.0), (4000, 7, 0.5), (5999, 3000, 3.0)):
        v = (rows[src] * sc).astype(np.float32)
        h.insert(ids[r].tobytes(), v); o.insert(ids[r].tobytes(), v)
    check("after in-place upserts")
    h.insert_batch(ids[n:], rows[n:]); o.insert_batch(ids[n:], rows[n:])
    check("after appends")
    for r in range(0, 3000, 7):
        h.remove(ids[r].tobytes()); o.remove(ids[r].tobytes())
    check("after removes")
    h.rebuild(); o.rebuild()
    check("after rebuild")
    v = (rows[11] * 1.5).astype(np.float32)
    h.insert(ids[6100].tobytes(), v); o.insert(ids[6100].tobytes(), v)
    check("upsert after rebuild")


@pytest.mark.parametrize("k,nq", [(10, 64), (100, 40)])
def test_search_batch_is_deterministic(hip, oracle, k, nq):
    """The batched kernel filters with bounds that depend on timing (list compactions, the cross-block slots); the
    bounds only ever drop rows that cannot be in the top k, so the RESULT must not move: twenty runs, bit for bit."""
    n, d = 200_000, 384
    rows = oracle.synth_rows(n, d)
    qs = oracle.synth_queries(n, d, nq)
    h = hip.HipIndex(d); h.insert_batch(ids_for(n), rows)
    ref = h.search_batch_arrays(qs, k)
    for _ in range(20):
        got = h.search_batch_arrays(qs, k)
        for a, b in zip(ref, got):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("d", [768, 384])
def test_full_size_headline_corpus(hip, oracle, d):
    """BASELINE's headline size, 1M x 768 f32, and config 2's, 1M x 384 (generated in HBM): the oracle's exact answers for a handful of queries,
    and size-independent properties for more — a row finds itself first with score 1, lists are ordered and hold
    distinct ids, the batched path agrees with the single-query path, and searching the two halves of the corpus
    separately and merging by (score, row) gives the whole-corpus answer."""
    import torch
    from cortex_amd import _lib
    L = _lib.load()
    n, k = 1_000_000, 10
    gen = torch.empty((n, d), dtype=torch.float32, device="cuda:0")
    assert L.cx_synth_fill_dev(0, gen.data_ptr(), oracle.SEED_CORPUS, oracle.SEED_CORPUS, oracle.SEED_DUP, n // 50, 0, n, d, 1) == 0
    ids = ids_for(n)
    h = hip.HipIndex(d)
    h.insert_batch_dev(ids, gen.data_ptr(), n, d)
    lut = {ids[i].tobytes(): i for i in range(n)}
    rng = np.random.default_rng(3)
    probe = np.sort(rng.choice(n, 64, replace=False))
    q_self = gen[torch.as_tensor(probe, device="cuda:0")].cpu().numpy()
    q_new = oracle.synth_queries(n, d, 6)

    # properties on 64 corpus rows used as queries
    single = []
    for p, q in zip(probe, q_self):
        gi, gs, gd = h.search_arrays(q, k)
        r = np.array([lut[x.tobytes()] for x in gi])
        assert len(r) == k and len(set(r.tolist())) == k and np.all(np.diff(gs) <= 0)
        assert gs[0] >= 1.0 - SCORE_TOL and (r[0] == p or gs[list(r).index(p)] >= 1.0 - SCORE_TOL)   # itself (or an exact duplicate with a lower row)
        ties = gs[:-1] == gs[1:]
        assert np.all(r[:-1][ties] < r[1:][ties])      # equal scores: insertion row ascending
        single.append((r, gs))
    bi, bs, bd, bc = h.search_batch_arrays(q_self, k)
    for j in range(64):
        assert int(bc[j]) == k
        assert_topk_parity(np.array([lut[x.tobytes()] for x in bi[j]]), bs[j], single[j][0], single[j][1], what=f"batch vs single q{j}")

    # two half-corpus indexes, merged by (score desc, row asc) == the whole corpus
    half = n // 2
    ha, hb = hip.HipIndex(d), hip.HipIndex(d)
    ha.insert_batch_dev(ids[:half], gen.data_ptr(), half, d)
    hb.insert_batch_dev(ids[half:], gen.data_ptr() + half * d * 4, n - half, d)
    for j in range(8):
        a = ha.search_arrays(q_self[j], k)
        b = hb.search_arrays(q_self[j], k)
        cand = [(-float(s), lut[i.tobytes()]) for i, s in zip(list(a[0]) + list(b[0]), list(a[1]) + list(b[1]))]
        cand.sort()
        assert_topk_parity(np.array([c[1] for c in cand[:k]]), np.array([-c[0] for c in cand[:k]]), single[j][0], single[j][1],
                           what=f"halves q{j}")
    del ha, hb

    # the oracle's exact brute force over the same 1M rows (3 GB on the host), held-out queries
    rows_h = gen.cpu().numpy()
    del gen
    o = oracle.OracleIndex(d)
    o.insert_batch(ids, rows_h)
    exp = o.search_batch(q_new, k, n_threads=6)
    for j, q in enumerate(q_new):
        gi, gs, gd = h.search_arrays(q, k)
        assert_topk_parity(np.array([lut[x.tobytes()] for x in gi]), gs, exp[j]["row"], exp[j]["score"], what=f"1M oracle q{j}")
        assert np.max(np.abs(gs - exp[j]["score"])) <= SCORE_TOL


@pytest.mark.parametrize("d,k", [(384, 10), (768, 40), (384, 100)])
def test_search_batch_zero_norm_rows_and_queries(hip, oracle, d, k):
    """NaN scores (zero-norm rows, zero queries: vector/index.rs:173-176 divides by both norms) through the batched
    MFMA path: NaN sorts after every number, in row order, exactly like the single-query path and the oracle."""
    n = 3000
    rows = oracle.synth_rows(n, d).copy()
    zero_rows = [3, 500, 2999]
    for r in zero_rows:
        rows[r] = 0.0
    h, o, ids = build_both(hip, oracle, rows)
    qs = oracle.synth_queries(n, d, 6).copy()
    qs[2] = 0.0                                     # a zero query: every score is NaN, results in row order
    bi, bs, bd, bc = h.search_batch_arrays(qs, k)
    for i in range(len(qs)):
        m = int(bc[i])
        e = o.search(qs[i], k)
        assert m == len(e["row"]) == k
        got = rows_of(ids, bi[i, :m])
        if i == 2:
            assert np.all(np.isnan(bs[i, :m])) and list(got) == list(range(k))
        else:
            assert not np.any(np.isnan(bs[i, :m]))  # 2997 real scores before any NaN
            assert_topk_parity(got, bs[i, :m], e["row"], e["score"], what=f"zero rows d={d} k={k} q{i}")
    # a corpus small enough that NaN rows reach the result: k = all rows
    small = rows[:40].copy()
    small[10] = 0.0
    small[3] = 0.0
    hs, os_, ids_s = build_both(hip, oracle, small)
    bi, bs, bd, bc = hs.search_batch_arrays(qs[:4], 40)
    for i in range(4):
        e = os_.search(qs[i], 40)
        got = rows_of(ids_s, bi[i, :int(bc[i])])
        assert int(bc[i]) == 40
        if i == 2:
            assert list(got) == list(range(40))
        else:
            assert list(got[-2:]) == [3, 10] and np.all(np.isnan(bs[i, 38:40])) and not np.any(np.isnan(bs[i, :38]))
            assert_topk_parity(got[:38], bs[i, :38], e["row"][:38], e["score"][:38], what=f"small d={d} q{i}")


@pytest.mark.parametrize("d,k,nq", [(384, 100, 90), (384, 50, 64), (768, 100, 40), (384, 10, 130)])
def test_search_batch_modes_with_filter_and_tombstones(hip, oracle, d, k, nq):
    """The same filter / tombstone semantics through every batched kernel mode: lists of 80 (k <= 32), wide lists for
    32 queries (768-d) and wide lists for all 64 queries (384-d, more than 32 queries in the call)."""
    n = 24000
    rows = oracle.synth_rows(n, d)
    qs = oracle.synth_queries(n, d, nq)
    h, o, ids = build_both(hip, oracle, rows)
    for r in range(0, n, 3):
        kind, agent = ("fact" if r % 2 else "event"), ("kai" if r % 5 else "rex")
        h.set_metadata(ids[r].tobytes(), kind, agent)
        o.set_metadata(ids[r].tobytes(), kind, agent)
    for r in list(range(100, 160)) + [0, n - 1]:
        h.remove(ids[r].tobytes()); o.remove(ids[r].tobytes())
    excl = [ids[i].tobytes() for i in range(1, 40)]
    hf, of = hip.VectorFilter(kinds=["fact"], exclude=excl, source_agent="kai"), oracle.Filter(kinds=["fact"], exclude=excl, source_agent="kai")
    bi, bs, bd, bc = h.search_batch_arrays(qs, k, hf)
    for i in range(nq):
        e = o.search(qs[i], k, of)
        m = int(bc[i])
        assert m == len(e["row"])
        assert_topk_parity(rows_of(ids, bi[i, :m]), bs[i, :m], e["row"], e["score"], what=f"modes d={d} k={k} q{i}")
    # and twice the same call gives the same bytes
    b2 = h.search_batch_arrays(qs, k, hf)
    assert np.array_equal(bi, b2[0]) and np.array_equal(bs, b2[1]) and np.array_equal(bc, b2[3])


def test_more_than_4G_elements(hip, oracle):
    """5.7M x 768 = 4.4e9 elements (17.5 GB): element offsets no longer fit 32 bits.  Rows near the end of the store
    must find themselves through the single-query scan, the batched kernel (split store) and the wide lists, and the
    oracle's exact answer over the last rows' neighbourhood must match."""
    import torch
    from cortex_amd import _lib
    L = _lib.load()
    n, d = 5_700_000, 768
    h = hip.HipIndex(d)
    h.reserve(n)
    chunk = 950_000
    keep = {}
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        gen = torch.empty((m, d), dtype=torch.float32, device="cuda:0")
        assert L.cx_synth_fill_dev(0, gen.data_ptr(), oracle.SEED_CORPUS, oracle.SEED_CORPUS, oracle.SEED_DUP, n // 50, lo, m, d, 1) == 0
        ids = np.zeros((m, 16), np.uint8)
        ids[:, 8:] = (np.arange(m, dtype=np.uint64) + np.uint64(lo)).astype(">u8").view(np.uint8).reshape(m, 8)
        h.insert_batch_dev(ids, gen.data_ptr(), m, d)
        for r in (0, m // 2, m - 1):
            keep[lo + r] = gen[r].cpu().numpy()
        del gen
    assert h.len() == n
    probes = sorted(keep)
    q = np.stack([keep[r] for r in probes])

    def row_of(id16):
        return int(np.frombuffer(id16.tobytes()[8:], dtype=">u8")[0])

    for r, v in zip(probes, q):
        gi, gs, gd = h.search_arrays(v, 5)
        assert gs[0] >= 1.0 - SCORE_TOL and r in [row_of(x) for x in gi[:2]], f"row {r} (single)"
        assert np.all(np.diff(gs) <= 0)
    for k in (10, 100):
        bi, bs, bd, bc = h.search_batch_arrays(np.concatenate([q] * 3)[:40], k)     # 40 queries: batched / wide (64-query) kernels
        for j in range(40):
            r = probes[j % len(probes)]
            assert int(bc[j]) == k and bs[j, 0] >= 1.0 - SCORE_TOL and r in [row_of(x) for x in bi[j, :2]], f"row {r} (batch k={k})"
            gi, gs, gd = h.search_arrays(q[j % len(probes)], k)
            assert_topk_parity(np.array([row_of(x) for x in bi[j, :k]]), bs[j, :k], np.array([row_of(x) for x in gi]), gs, what=f"batch vs single k={k} q{j}")


@pytest.mark.parametrize("d", [384, 768, 1024, 200])
def test_scores_against_f64_ground_truth(hip, oracle, d):
    """Neither restatement is the judge of the arithmetic here: the exact cosine in float64 is.  Single-query scan,
    batched MFMA search (bf16 hi/lo split, three products) and the threshold path must all report
    clamp(cos, 0, 1) within SCORE_TOL of it (measured: ~1e-6), and their top-k must be the f64 top-k up to near-ties."""
    n, nq = 30_000, 64
    rows = oracle.synth_rows(n, d)
    qs = oracle.synth_queries(n, d, nq)
    qs[1] *= 3.5                                       # un-normalised query: the division by |q| is part of the answer
    R, Q = rows.astype(np.float64), qs.astype(np.float64)
    cos = (Q @ R.T) / (np.linalg.norm(Q, axis=1)[:, None] * np.linalg.norm(R, axis=1)[None, :])
    truth = np.clip(cos, 0.0, 1.0)
    ids = ids_for(n)
    h = hip.HipIndex(d); h.insert_batch(ids, rows)
    row_of = lambda gi: gi[:, 8:].copy().view(">u8").reshape(-1).astype(np.int64)
    worst = 0.0
    for k in (10, 100):
        bi, bs, bd, bc = h.search_batch_arrays(qs, k)
        for j in range(nq):
            order = np.lexsort((np.arange(n), -truth[j]))[:k]
            paths = {"batch": (row_of(bi[j][:int(bc[j])]), bs[j][:int(bc[j])], bd[j][:int(bc[j])])}
            if j < 16:
                gi, gs, gd = h.search_arrays(qs[j], k)
                paths["single"] = (row_of(gi), gs, gd)
            for name, (r, sc, di) in paths.items():
                assert len(r) == k
                err = np.abs(sc.astype(np.float64) - truth[j, r])
                worst = max(worst, float(err.max()))
                assert err.max() <= SCORE_TOL, f"{name} d={d} k={k} q{j}: |score - f64| = {err.max()}"
                assert np.max(np.abs(di.astype(np.float64) - (1.0 - cos[j, r]))) <= SCORE_TOL
                assert_topk_parity(r, sc, order, truth[j, order], what=f"{name} d={d} k={k} q{j} vs f64")
    gi, gs, gd = h.search_threshold_arrays(qs[0], 0.6)
    r = row_of(gi)
    assert np.max(np.abs(gs.astype(np.float64) - truth[0, r])) <= SCORE_TOL
    inside = set(np.nonzero(truth[0] >= 0.6 + SCORE_TOL)[0].tolist())
    outside = set(np.nonzero(truth[0] < 0.6 - SCORE_TOL)[0].tolist())
    assert inside <= set(r.tolist()) and not (outside & set(r.tolist()))
    assert worst <= 1e-5, f"scores drifted from the f64 truth: {worst}"


@pytest.mark.gpu
@pytest.mark.parametrize("env", [
    {"CX_BATCH2": "0"},                                                   # 768-d through batchg.hip (batch2_kernel normally serves it)
    {"CX_BATCHG_FILTER_MIN": "1000"},                                     # the bound + candidates path at test sizes
    {"CX_BATCHG_FILTER_MIN": "1000", "CX_BATCHG_SAMPLE_STEP": "100000"},  # a one-tile sample: weak bounds, lists overflow, exact fallback
    {"CX_BATCHG_FILTER": "0"},                                            # the dense pass only
])
def test_search_batch_other_kernel_instances(env):
    """batchg.hip's paths the default dispatch does not reach at test sizes — each against the oracle in its own
    process (the switches are read once)."""
    import subprocess, sys, os
    code = r"""
import numpy as np, sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import cortex_amd
from oracle import oracle
from conftest import assert_topk_parity, ids_for
oracle.build()
for n, d, k, nq in [(3001, 768, 10, 70), (1520, 768, 100, 9), (9100, 1024, 10, 64), (30000, 512, 32, 7), (130, 256, 5, 40), (5000, 1536, 256, 3)]:
    rows = oracle.synth_rows(n, d); qs = oracle.synth_queries(n, d, nq); ids = ids_for(n)
    lut = {ids[i].tobytes(): i for i in range(n)}
    h = cortex_amd.HipIndex(d); h.insert_batch(ids, rows)
    o = oracle.OracleIndex(d); o.insert_batch(ids, rows)
    bi, bs, bd, bc = h.search_batch_arrays(qs, k)
    for i in range(nq):
        e = o.search(qs[i], k); m = int(bc[i])
        assert m == len(e["row"]), (m, len(e["row"]))
        got = np.array([lut[g.tobytes()] for g in bi[i, :m]], dtype=np.int64)
        assert_topk_parity(got, bs[i, :m], e["row"], e["score"], what="n=%%d d=%%d k=%%d q%%d" %% (n, d, k, i))
    if n == 9100:   # the same store with removed rows, metadata and a filter: the bound counts passing rows only
        for r in range(0, n, 3):
            kind = "fact" if r %% 2 else "event"
            h.set_metadata(ids[r].tobytes(), kind, "kai"); o.set_metadata(ids[r].tobytes(), kind, "kai")
        for r in (10, 11, 500, 4097):
            h.remove(ids[r].tobytes()); o.remove(ids[r].tobytes())
        excl = [ids[i].tobytes() for i in (1, 2, 3, 130)]
        for hf, of in ((None, None), (cortex_amd.VectorFilter(kinds=["fact"], exclude=excl), oracle.Filter(kinds=["fact"], exclude=excl))):
            bi, bs, bd, bc = h.search_batch_arrays(qs, k, hf)
            for i in range(nq):
                e = o.search(qs[i], k, of); m = int(bc[i])
                assert m == len(e["row"]), (m, len(e["row"]))
                got = np.array([lut[g.tobytes()] for g in bi[i, :m]], dtype=np.int64)
                assert_topk_parity(got, bs[i, :m], e["row"], e["score"], what="filtered n=%%d d=%%d k=%%d q%%d" %% (n, d, k, i))
# fewer than k rows with a positive cosine: the k-th best score is the clamped 0.0, and the rows tied at 0 must come out in the
# declared order (row ascending) whichever path ran — batch == single scan == oracle, id for id (round-2 ADVICE, batchg.hip)
n, d, k = 4000, 1024, 10
rng = np.random.default_rng(8)
rows = np.abs(rng.normal(size=(n, d))).astype(np.float32)            # the positive orthant
qs = -np.abs(rng.normal(size=(8, d))).astype(np.float32)             # cosine < 0 with every row ...
for t in range(6): rows[100 + 37 * t] = -rows[100 + 37 * t]          # ... but six
ids = ids_for(n); lut = {ids[i].tobytes(): i for i in range(n)}
h = cortex_amd.HipIndex(d); h.insert_batch(ids, rows)
o = oracle.OracleIndex(d); o.insert_batch(ids, rows)
bi, bs, bd, bc = h.search_batch_arrays(qs, k)
for i in range(len(qs)):
    e = o.search(qs[i], k); si, ss, sd = h.search_arrays(qs[i], k)
    got = [lut[g.tobytes()] for g in bi[i, :int(bc[i])]]
    assert got == [lut[g.tobytes()] for g in si] == [int(r) for r in e["row"]], (i, got, e["row"])
    assert int((bs[i, :k] > 0).sum()) == 6
print("ok")
""" % ((os.path.dirname(os.path.dirname(os.path.abspath(__file__))),) * 2)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
@pytest.mark.parametrize("d,dtype", [(768, "f32"), (384, "f32"), (1024, "bf16"), (128, "f32"), (1024, "f32"), (640, "bf16")])
def test_search_batch_screening_kernel(d, dtype):
    """batchs.hip (stores of >= 131,072 rows by default: a bf16 screening pass with a rigorous error bound, survivors
    re-scored exactly) at test sizes, against the oracle in its own process: ragged tiles, every k class, several passes,
    removed rows + metadata + filter, zero rows / zero query, massive ties, fewer tiles than slots, fewer rows with a
    positive cosine than k.  A bf16 store is compared with the oracle on the rounded rows."""
    import subprocess, sys, os
    code = r"""
import numpy as np, sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import cortex_amd
from oracle import oracle
from conftest import assert_topk_parity, ids_for
oracle.build()
d, dtype = int(sys.argv[1]), sys.argv[2]
def rnd(x):
    if dtype != "bf16": return x
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(x.shape)
def both(rows):
    ids = ids_for(len(rows)); lut = {ids[i].tobytes(): i for i in range(len(rows))}
    h = cortex_amd.HipIndex(d, dtype=dtype); h.insert_batch(ids, rows)
    o = oracle.OracleIndex(d); o.insert_batch(ids, rnd(rows))
    return h, o, ids, lut
def check(h, o, lut, qs, k, hf=None, of=None, what="", exact_ids=False):
    bi, bs, bd, bc = h.search_batch_arrays(qs, k, hf)
    for i in range(len(qs)):
        e = o.search(qs[i], k, of); m = int(bc[i])
        assert m == len(e["row"]), (what, i, m, len(e["row"]))
        got = np.array([lut[g.tobytes()] for g in bi[i, :m]], dtype=np.int64)
        if exact_ids:
            assert list(map(int, got)) == list(map(int, e["row"])), (what, i, got[:8], e["row"][:8])
        else:
            assert_topk_parity(got, bs[i, :m], e["row"], e["score"], what="%%s q%%d" %% (what, i))
# (45000 rows, 128 / 97 / 130 queries: at row widths up to 512 a call of more than 64 queries runs 128 per pass — two banks)
for n, k, nq in [(3001, 10, 100), (40000, 100, 70), (20000, 256, 5), (777, 32, 7), (300, 100, 64), (257, 1, 3), (60000, 10, 64), (45000, 10, 128), (45000, 32, 97), (45000, 100, 130)]:
    if d >= 1024 and n > 20000: n = 20000
    rows = oracle.synth_rows(n, d); qs = oracle.synth_queries(n, d, nq)
    h, o, ids, lut = both(rows)
    check(h, o, lut, qs, k, what="n=%%d k=%%d" %% (n, k))
    if k == 10 and nq == 64:
        for r in range(0, n, 3):
            kind = "fact" if r %% 2 else "event"
            h.set_metadata(ids[r].tobytes(), kind, "kai"); o.set_metadata(ids[r].tobytes(), kind, "kai")
        for r in (10, 11, 500, 4097, n - 1):
            h.remove(ids[r].tobytes()); o.remove(ids[r].tobytes())
        excl = [ids[i].tobytes() for i in (1, 2, 3, 130)]
        check(h, o, lut, qs, k, what="tombstones")
        check(h, o, lut, qs, k, cortex_amd.VectorFilter(kinds=["fact"], exclude=excl), oracle.Filter(kinds=["fact"], exclude=excl), what="filtered")
        for r in range(0, 40): h.set_metadata(ids[7 * r + 1].tobytes(), "rare", "zed"); o.set_metadata(ids[7 * r + 1].tobytes(), "rare", "zed")
        check(h, o, lut, qs, 50, cortex_amd.VectorFilter(kinds=["rare"]), oracle.Filter(kinds=["rare"]), what="rare kind")
        # every row with metadata: a filter passing 1 row in 500 (bounds from a handful of rows say little or never form: the workers
        # look at the rows' metadata themselves then) and one passing 1 in 10
        kinds_all = ["sparse" if r %% 500 == 7 else ("tenth" if r %% 10 == 3 else "common") for r in range(n)]
        h.set_metadata_batch(ids, kinds_all, ["kai"] * n)
        for r in range(n): o.set_metadata(ids[r].tobytes(), kinds_all[r], "kai")
        for kk in (10, 50):
            check(h, o, lut, qs, kk, cortex_amd.VectorFilter(kinds=["sparse"]), oracle.Filter(kinds=["sparse"]), what="1 in 500 passes k=%%d" %% kk)
            check(h, o, lut, qs, kk, cortex_amd.VectorFilter(kinds=["tenth"]), oracle.Filter(kinds=["tenth"]), what="1 in 10 passes k=%%d" %% kk)
        check(h, o, lut, qs, 10, cortex_amd.VectorFilter(kinds=["sparse", "tenth"], exclude=excl), oracle.Filter(kinds=["sparse", "tenth"], exclude=excl), what="two kinds")
        # in-place upserts and appends after the screening copy exists
        for r, src, sc in ((5, 100, 2.0), (4000, 7, 0.5)):
            v = (rows[src] * sc).astype(np.float32)
            h.insert(ids[r].tobytes(), v); o.insert(ids[r].tobytes(), rnd(v))
        check(h, o, lut, qs, k, what="after upserts")
# zero rows and a zero query: NaN scores come last, in row order
n = 5000
rows = oracle.synth_rows(n, d); rows[[3, 77, 4000]] = 0.0
qs = oracle.synth_queries(n, d, 9); qs[4] = 0.0
h, o, ids, lut = both(rows)
for k in (10, 100): check(h, o, lut, qs, k, what="zeros k=%%d" %% k)
# 50 distinct vectors repeated: almost every cut falls inside a run of equal scores; ids exact
n = 20000
base = oracle.synth_rows(50, d)
rows = np.ascontiguousarray(base[np.arange(n) %% 50])
qs = np.ascontiguousarray(base[:16] + 0.05 * oracle.synth_queries(50, d, 16))
h, o, ids, lut = both(rows)
for k in (32, 100): check(h, o, lut, qs, k, what="ties k=%%d" %% k, exact_ids=True)
# fewer than k rows with a positive cosine: rows tied at the clamped 0.0 come out in row order
n, k = 4000, 10
rng = np.random.default_rng(8)
rows = np.abs(rng.normal(size=(n, d))).astype(np.float32)
qs = -np.abs(rng.normal(size=(8, d))).astype(np.float32)
for t in range(6): rows[100 + 37 * t] = -rows[100 + 37 * t]
h, o, ids, lut = both(rows)
check(h, o, lut, qs, k, what="six positive", exact_ids=True)
# rows in ASCENDING order of similarity to the queries (all near one direction): every tile raises the bound, so nearly every
# candidate came in under a bound far below the pass's last one — the re-score's second look strikes most of the list, and
# what it keeps must still hold the exact k best (and the reverse order: the first tiles already hold them)
n = 30000 if d < 1024 else 12000
rng = np.random.default_rng(21)
c = rng.normal(size=d).astype(np.float32); c /= np.linalg.norm(c)
noise = rng.normal(size=(n, d)).astype(np.float32); noise /= np.linalg.norm(noise, axis=1, keepdims=True)
w = np.linspace(0.0, 1.0, n, dtype=np.float32)[:, None]
asc = (w * c[None, :] + (1.0 - w) * noise).astype(np.float32)
qs = (c[None, :] + 0.1 * rng.normal(size=(20, d))).astype(np.float32)
for name, rows in (("ascending", asc), ("descending", np.ascontiguousarray(asc[::-1]))):
    h, o, ids, lut = both(rows)
    for k in (10, 100): check(h, o, lut, qs, k, what="%%s k=%%d" %% (name, k))
print("ok")
""" % ((os.path.dirname(os.path.dirname(os.path.abspath(__file__))),) * 2)
    # (CX_BATCHS_LOC_MIN=0 at three of the six widths: the blocks' local first bounds — which large passes only use by default —
    # with hits held in the rings until the grid's bounds arrive)
    env = dict(os.environ, CX_BATCHS_MIN_ROWS="256")
    if (d, dtype) in ((768, "f32"), (1024, "bf16"), (128, "f32")):
        env["CX_BATCHS_LOC_MIN"] = "0"
    r = subprocess.run([sys.executable, "-c", code, str(d), dtype], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
def test_batched_search_bound_and_candidates_at_scale(hip):
    """400k x 1024 (1.6 GB), 70 queries, k = 10 and 100: batchg.hip's default path at this size — a bound per query from a
    1-in-32 sample of the row tiles, then only the rows that reach it — must return what 70 single-query scans return
    (ids, order, scores: both paths compute f32-exact cosines, the scan is pinned to the oracle above)."""
    import torch
    from cortex_amd import _lib
    L = _lib.load()
    n, d, nq = 400_000, 1024, 70
    gen = torch.empty((n, d), dtype=torch.float32, device="cuda:0")
    assert L.cx_synth_fill_dev(0, gen.data_ptr(), 20260313, 20260313, 20260315, n // 50, 0, n, d, 1) == 0
    ids = np.zeros((n, 16), np.uint8)
    ids[:, 8:] = np.arange(n, dtype=np.uint64).astype(">u8").view(np.uint8).reshape(n, 8)
    h = hip.HipIndex(d)
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
            assert_topk_parity(got, bs[i, :m], exp, gs, what=f"k={k} q{i}")
