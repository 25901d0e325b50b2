"""The auto-linker's batched similarity pass (bf16 MFMA filter + exact f32 rescore + link rules)
against the oracle's restatement of linker/auto_linker.rs:215-264 + linker/rules.rs:42-62 and
linker/dedup.rs:65-127.

Bar: the edge list (from, to) is identical, in order, to the oracle's, and weights agree within
SCORE_TOL, except that a pair whose oracle score lies within SCORE_TOL of the threshold may fall
on either side (and then shifts what a per-node cap admits): such rows are compared as sets of
near-threshold-tolerant prefixes."""
import numpy as np
import pytest

from conftest import SCORE_TOL, assert_topk_parity, ids_for

pytestmark = pytest.mark.gpu


def per_node(fr, to, w):
    out = {}
    for a, b, s in zip(fr, to, w):
        out.setdefault(int(a), []).append((int(b), float(s)))
    return out


def compare_edges(got, exp, thr, scores_of, what):
    """got/exp: dict from_row -> [(to_row, weight)] in order."""
    assert set(got) | set(exp) == set(exp) | set(got)
    for node in sorted(set(got) | set(exp)):
        g, e = got.get(node, []), exp.get(node, [])
        if [x[0] for x in g] == [x[0] for x in e]:
            for (_, a), (_, b) in zip(g, e):
                assert abs(a - b) <= SCORE_TOL, f"{what}: node {node} weight {a} vs {b}"
            continue
        # differences must be explained by near-ties or near-threshold scores
        sc = scores_of(node)
        gs, es = set(x[0] for x in g), set(x[0] for x in e)
        for j in gs ^ es:
            near_thr = abs(sc[j] - thr) <= SCORE_TOL
            near_tail = e and abs(sc[j] - e[-1][1]) <= SCORE_TOL
            assert near_thr or near_tail, f"{what}: node {node} neighbour {j} score {sc[j]} unexplained"
        common = [x for x in g if x[0] in es]
        for (j, a) in common:
            assert abs(a - sc[j]) <= SCORE_TOL


def build(hip, oracle, rows):
    ids = ids_for(len(rows))
    h = hip.HipIndex(rows.shape[1])
    h.insert_batch(ids, rows)
    o = oracle.OracleIndex(rows.shape[1])
    o.insert_batch(ids, rows)
    return h, o, ids


def oracle_scores(o, rows):
    def f(node):
        r = o.search(rows[node], len(rows))
        out = np.zeros(len(rows))
        out[r["row"]] = r["score"]
        return out
    return f


@pytest.mark.parametrize("n,d,thr,topk,cap", [
    (3000, 768, 0.85, 100, 50),    # BASELINE config 3 shape, reduced rows
    (3000, 768, 0.75, 100, 50),    # the reference's default threshold
    (2000, 384, 0.75, 100, 50),    # BGE-small dim
    (1500, 1024, 0.85, 20, 5),     # small k and cap
    (700, 100, 0.75, 100, 50),     # dim % 64 != 0 -> exact scan path for every row
    (130, 64, 0.5, 100, 50),       # fewer rows than one tile
])
def test_autolink_pass_matches_oracle(hip, oracle, n, d, thr, topk, cap):
    rows = oracle.synth_rows(n, d)
    h, o, ids = build(hip, oracle, rows)
    thr32 = float(np.float32(thr))
    fr, to, w = h.autolink_pass_rows(None, topk, thr32, cap)
    e = o.autolink_pass(np.arange(n), topk, thr32, cap, n_threads=8)
    got, exp = per_node(fr, to, w), per_node(e["from_row"], e["to_row"], e["weight"])
    assert len(exp) > 0
    compare_edges(got, exp, thr32, oracle_scores(o, rows), f"n={n} d={d} thr={thr}")
    # scan order then score order
    assert np.all(np.diff(fr.astype(np.int64)) >= 0)
    for node, lst in got.items():
        assert all(lst[i][1] >= lst[i + 1][1] for i in range(len(lst) - 1))
        assert len(lst) <= cap and all(j != node for j, _ in lst)


def test_scan_subset_order_and_deleted_neighbours(hip, oracle):
    n, d, thr = 2500, 768, float(np.float32(0.75))
    rows = oracle.synth_rows(n, d)
    h, o, ids = build(hip, oracle, rows)
    rng = np.random.default_rng(7)
    scan = rng.permutation(n)[:300].astype(np.uint32)          # arbitrary order, subset (the 500-node batch)
    deleted = (rng.random(n) < 0.1).astype(np.uint8)            # storage tombstones still indexed (Q2)
    fr, to, w = h.autolink_pass_rows(scan, 100, thr, 50, deleted)
    e = o.autolink_pass(scan, 100, thr, 50, deleted, n_threads=8)
    assert not np.any(deleted[to])
    # scan order preserved
    order = {int(r): i for i, r in enumerate(scan)}
    pos = np.array([order[int(a)] for a in fr])
    assert np.all(np.diff(pos) >= 0)
    compare_edges(per_node(fr, to, w), per_node(e["from_row"], e["to_row"], e["weight"]), thr,
                  oracle_scores(o, rows), "subset+deleted")


def _csr(lists):
    off = np.zeros(len(lists) + 1, np.uint64)
    off[1:] = np.cumsum([len(x) for x in lists])
    return off, np.array([t for x in lists for t in x], dtype=np.uint32)


@pytest.mark.parametrize("n,d,scan_all", [(3000, 768, True), (2500, 384, False), (700, 100, True)])
def test_rescan_with_existing_edges_matches_oracle(hip, oracle, n, d, scan_all):
    """The rescan after a threshold/model change (auto_linker.rs:137-182): ~30 % of the scanned nodes already have
    outgoing related_to edges.  The reference drops those WITHOUT counting them towards max_edges_per_node
    (:226-231, :249-258) and keeps walking its top-100 list; then take(max_edges_per_cycle) (:284-287).
    The fused pass must agree with the oracle edge for edge."""
    rows = oracle.synth_rows(n, d)
    h, o, ids = build(hip, oracle, rows)
    rng = np.random.default_rng(11)
    thr_old, thr_new, cap = float(np.float32(0.85)), float(np.float32(0.75)), 10
    scan = None if scan_all else rng.permutation(n)[:600].astype(np.uint32)
    scan_o = np.arange(n, dtype=np.uint32) if scan is None else scan
    first = o.autolink_pass(scan_o, 100, thr_old, cap, n_threads=8)       # what an earlier cycle created
    have = {}
    for e in first:
        have.setdefault(int(e["from_row"]), []).append(int(e["to_row"]))
    lists, with_edges = [], 0
    for s in scan_o:
        mine = have.get(int(s), [])
        if mine and rng.random() < 0.3:
            keep = mine if rng.random() < 0.5 else mine[: max(1, len(mine) // 2)]
            extra = [int(x) for x in rng.integers(0, n, 3)]               # edges to unrelated nodes (other cycles' work)
            lists.append(list(rng.permutation(keep + extra)))            # unsorted on purpose
            with_edges += 1
        else:
            lists.append([])
    assert with_edges > 0.1 * len(scan_o) * (len(have) / len(scan_o))
    existing = _csr(lists)
    fr, to, w = h.autolink_pass_rows(scan, 100, thr_new, cap, existing=existing)
    e = o.autolink_pass(scan_o, 100, thr_new, cap, n_threads=8, existing=existing)
    got, exp = per_node(fr, to, w), per_node(e["from_row"], e["to_row"], e["weight"])
    compare_edges(got, exp, thr_new, oracle_scores(o, rows), f"rescan n={n} d={d}")
    # compare_edges tolerates near-ties (two neighbours whose scores differ by less than SCORE_TOL may swap and, at a
    # cap boundary, replace each other); they must stay the exception: nearly every node's list is identical, in order
    same = sum(1 for node in exp if [x[0] for x in got.get(node, [])] == [x[0] for x in exp[node]])
    assert same >= 0.98 * len(exp), f"only {same} of {len(exp)} nodes have the oracle's exact list"
    assert len(fr) == len(e)
    sets = {int(s): set(l) for s, l in zip(scan_o, lists)}
    assert all(int(b) not in sets[int(a)] for a, b in zip(fr, to))        # never proposed again
    # the pass without `existing` differs (it spends its cap on edges the node already has)
    fr0, to0, _ = h.autolink_pass_rows(scan, 100, thr_new, cap)
    assert list(zip(fr0.tolist(), to0.tolist())) != list(zip(fr.tolist(), to.tolist()))
    # per-cycle cap: the first max_edges_per_cycle proposals in scan order
    frc, toc, wc = h.autolink_pass_rows(scan, 100, thr_new, cap, existing=existing, max_edges_per_cycle=2000)
    m = min(2000, len(fr))
    assert len(frc) == m and np.array_equal(frc, fr[:m]) and np.array_equal(toc, to[:m]) and np.array_equal(wc, w[:m])
    ne, _ = h.autolink_pass_timed(100, thr_new, cap, scan, existing=existing, max_edges_per_cycle=2000)
    assert ne == m


def test_existing_edges_validation_and_cap_semantics(hip, oracle):
    n, d = 600, 128
    rows = oracle.synth_rows(n, d)
    h, o, ids = build(hip, oracle, rows)
    scan = np.array([9, 20, 33], dtype=np.uint32)
    with pytest.raises(hip.ValidationError):
        h.autolink_pass_rows(scan, 100, 0.5, 5, existing=(np.array([0, 1], np.uint64), np.array([1], np.uint32)))
    with pytest.raises(hip.ValidationError):                    # offsets must not decrease
        h.autolink_pass_rows(scan, 100, 0.5, 5, existing=(np.array([0, 2, 1, 2], np.uint64), np.array([1, 2], np.uint32)))
    # the cap is tested after the push (:259-262): 0 behaves like 1, as in the reference
    fr0, to0, w0 = h.autolink_pass_rows(scan, 100, 0.0, 0)
    fr1, to1, w1 = h.autolink_pass_rows(scan, 100, 0.0, 1)
    e0 = o.autolink_pass(scan, 100, np.float32(0.0), 0)
    assert np.array_equal(fr0, fr1) and np.array_equal(to0, to1) and len(fr0) == 3
    assert list(zip(fr0.tolist(), to0.tolist())) == list(zip(e0["from_row"].tolist(), e0["to_row"].tolist()))
    assert len(h.autolink_pass_rows(scan, 100, 0.0, 5, max_edges_per_cycle=0)[0]) == 0


def test_candidate_overflow_takes_the_exact_path(hip, oracle, monkeypatch):
    monkeypatch.setenv("CX_PAIR_CAND_CAP", "16")                # clusters hold ~50 rows: most lists overflow
    n, d, thr = 1500, 768, float(np.float32(0.75))
    rows = oracle.synth_rows(n, d)
    h, o, ids = build(hip, oracle, rows)
    fr, to, w = h.autolink_pass_rows(None, 100, thr, 50)
    e = o.autolink_pass(np.arange(n), 100, thr, 50, n_threads=8)
    compare_edges(per_node(fr, to, w), per_node(e["from_row"], e["to_row"], e["weight"]), thr,
                  oracle_scores(o, rows), "overflow")


def test_pass_sees_upserts_and_removes(hip, oracle):
    n, d, thr = 1200, 768, float(np.float32(0.8))
    rows = oracle.synth_rows(n, d)
    h, o, ids = build(hip, oracle, rows)
    h.autolink_pass_rows(None, 100, thr, 50)                    # builds the bf16 shadow
    new = oracle.synth_queries(n, d, 3)
    for i, r in enumerate((5, 600, 1100)):                      # in-place upserts -> stale shadow rows refreshed
        h.insert(ids[r].tobytes(), new[i]); o.insert(ids[r].tobytes(), new[i])
    h.remove(ids[7].tobytes()); o.remove(ids[7].tobytes())      # removed rows never come back as neighbours
    extra = oracle.synth_rows(n + 100, d, n, 100)               # appended rows extend the shadow
    eids = ids_for(n + 100)[n:]
    h.insert_batch(eids, extra); o.insert_batch(eids, extra)
    allrows = np.concatenate([rows, extra]); allrows[[5, 600, 1100]] = new
    scan = np.array([r for r in range(n + 100) if r != 7], dtype=np.uint32)
    fr, to, w = h.autolink_pass_rows(scan, 100, thr, 50)
    e = o.autolink_pass(scan, 100, thr, 50, n_threads=8)
    assert 7 not in set(to.tolist())
    o2 = oracle.OracleIndex(d); o2.insert_batch(ids_for(n + 100), allrows)
    compare_edges(per_node(fr, to, w), per_node(e["from_row"], e["to_row"], e["weight"]), thr,
                  oracle_scores(o2, allrows), "after mutations")


@pytest.mark.parametrize("n,d,thr", [(1800, 768, 0.92), (1500, 384, 0.92), (400, 100, 0.9)])
def test_dedup_scan_matches_oracle(hip, oracle, n, d, thr):
    rows = oracle.synth_rows(n, d)
    h, o, ids = build(hip, oracle, rows)
    rng = np.random.default_rng(3)
    for deleted in (None, (rng.random(n) < 0.05).astype(np.uint8)):
        a, b, s = h.dedup_scan_rows(float(np.float32(thr)), deleted)
        e = o.dedup_scan(float(np.float32(thr)), deleted)
        got = {(int(x), int(y)): float(z) for x, y, z in zip(a, b, s)}
        exp = {(int(x), int(y)): float(z) for x, y, z in zip(e["from_row"], e["to_row"], e["weight"])}
        assert len(exp) > 0
        for p in set(got) ^ set(exp):
            sc = got.get(p, exp.get(p))
            assert abs(sc - np.float32(thr)) <= SCORE_TOL, f"pair {p} score {sc} not a threshold tie"
        for p in set(got) & set(exp):
            assert abs(got[p] - exp[p]) <= SCORE_TOL
        assert np.all(np.diff(a.astype(np.int64)) >= 0)          # scan (row) order
        if deleted is not None:
            assert not np.any(deleted[a])                         # deleted nodes are never the scanning side


def _with_duplicate_cluster(oracle, n, d, members, seed=21):
    """synthetic rows with `members` near-copies of one vector spread over the store (a re-imported corpus, boilerplate nodes)"""
    rows = oracle.synth_rows(n, d).copy()
    rng = np.random.default_rng(seed)
    where = np.sort(rng.choice(n, size=members, replace=False))
    base = rows[where[0]].copy()
    rows[where] = base[None, :] + rng.normal(0.0, 0.02 / np.sqrt(d), size=(members, d)).astype(np.float32)
    rows[where[5]] = base                      # and one exact duplicate
    return rows, where


@pytest.mark.parametrize("n,d,members", [(1600, 384, 400), (900, 768, 300)])
def test_dedup_scan_has_no_neighbour_cap(hip, oracle, n, d, members):
    """dedup.rs:85-87 calls search_threshold: a node with more near-duplicates than the pass's lists are wide (256) still
    reports every pair.  A cluster of 300-400 near-copies (~45k-80k pairs): pair for pair what cxo_dedup_scan reports."""
    rows, where = _with_duplicate_cluster(oracle, n, d, members)
    h, o, ids = build(hip, oracle, rows)
    thr = float(np.float32(0.92))
    rng = np.random.default_rng(4)
    for deleted in (None, (rng.random(n) < 0.05).astype(np.uint8)):
        a, b, s = h.dedup_scan_rows(thr, deleted)
        e = o.dedup_scan(thr, deleted)
        got = {(int(x), int(y)): float(z) for x, y, z in zip(a, b, s)}
        exp = {(int(x), int(y)): float(z) for x, y, z in zip(e["from_row"], e["to_row"], e["weight"])}
        assert len(exp) > members * (members - 1) // 2 * 0.8
        for p in set(got) ^ set(exp):
            sc = got.get(p, exp.get(p))
            assert abs(sc - np.float32(thr)) <= SCORE_TOL, f"pair {p} score {sc} not a threshold tie"
        for p in set(got) & set(exp):
            assert abs(got[p] - exp[p]) <= SCORE_TOL
        assert len(got) == len(a), "a pair was reported twice"
        assert np.all(np.diff(a.astype(np.int64)) >= 0)          # scan (row) order
        # inside one scanned row: best first, like search_threshold's list
        for node in where[:3]:
            sel = s[a == node]
            assert np.all(np.diff(sel) <= 0)


def test_linker_mirror_returns_reference_shaped_edges(hip, oracle):
    from cortex_amd import SimilarityConfig
    from cortex_amd.linker import autolink_similarity_edges, dedup_scan
    import uuid
    rows = oracle.synth_rows(1000, 768)
    ids = ids_for(1000)
    h = hip.HipIndex(768)
    h.insert_batch(ids, rows)
    cfg = SimilarityConfig.default()
    edges = autolink_similarity_edges(h, [ids[i].tobytes() for i in range(50)], cfg, max_edges_per_node=50)
    assert edges and all(e.relation == "related_to" and e.weight >= np.float32(0.75) for e in edges)
    assert all(e.provenance["AutoSimilarity"]["score"] == e.weight and e.from_id != e.to_id for e in edges)
    assert {e.from_id.bytes for e in edges} <= {ids[i].tobytes() for i in range(50)}
    pairs = dedup_scan(h, cfg)
    assert all(p.similarity >= np.float32(0.92) and isinstance(p.node_a, uuid.UUID) for p in pairs)


def test_full_size_pass_is_consistent_with_search(hip, oracle):
    """BASELINE config 3 size (100k x 768, thr 0.85): size-independent properties — every edge's weight
    is what search() reports for that pair, lists are ordered and capped, and for sampled nodes the
    edges equal the reference's walk over search(emb, 100)."""
    import torch
    from cortex_amd import _lib
    L = _lib.load()
    n, d, thr = 100_000, 768, float(np.float32(0.85))
    gen = torch.empty((n, d), dtype=torch.float32, device="cuda:0")
    assert L.cx_synth_fill_dev(0, gen.data_ptr(), oracle.SEED_CORPUS, oracle.SEED_CORPUS, oracle.SEED_DUP, n // 50, 0, n, d, 1) == 0
    h = hip.HipIndex(d)
    ids = ids_for(n)
    h.insert_batch_dev(ids, gen.data_ptr(), n, d)
    fr, to, w = h.autolink_pass_rows(None, 100, thr, 50)
    assert len(fr) > n  # clustered data: tens of neighbours above 0.85 per node
    got = per_node(fr, to, w)
    rows_h = gen.cpu().numpy()
    lut = {ids[i].tobytes(): i for i in range(n)}
    rng = np.random.default_rng(11)
    for node in rng.choice(n, 40, replace=False):
        gi, gs, gd = h.search_arrays(rows_h[node], 100)
        walk = []
        for i in range(len(gs)):
            j = lut[gi[i].tobytes()]
            if j == node:
                continue
            if gs[i] >= np.float32(thr):
                walk.append((j, float(gs[i])))
            if len(walk) >= 50:
                break
        g = got.get(int(node), [])
        assert [x[0] for x in g] == [x[0] for x in walk] or all(
            abs(s - thr) <= SCORE_TOL or abs(s - (walk[-1][1] if walk else thr)) <= SCORE_TOL
            for j, s in set(g) ^ set(walk)), f"node {node}"
        for (j1, s1), (j2, s2) in zip(g, walk):
            if j1 == j2:
                assert abs(s1 - s2) <= SCORE_TOL


def test_topk_lists_and_generic_walk(hip, oracle):
    """cx_topk_lists_rows = the reference's search(emb, 100, None) per scanned node; autolink_walk over those
    lists with the similarity rule alone reproduces the fused pass edge for edge, and with an always-firing
    structural rule added (SURVEY a14': legacy rules on) it matches the same walk over the ORACLE's lists."""
    from cortex_amd import linker
    n, d = 3000, 768
    rows = oracle.synth_rows(n, d)
    ids = ids_for(n)
    h = hip.HipIndex(d); h.insert_batch(ids, rows)
    o = oracle.OracleIndex(d); o.insert_batch(ids, rows)
    for r in (7, 1500):
        h.remove(ids[r].tobytes()); o.remove(ids[r].tobytes())
    scan = np.array([0, 5, 8, 999, 1501, 2999, 42, 43, 44, 45], dtype=np.uint32)
    scan_ids = [ids[i].tobytes() for i in scan]
    lr, ls, lc = h.topk_lists_rows(100, scan)
    for p, i in enumerate(scan):
        e = o.search(rows[i], 100)
        m = int(lc[p])
        assert m == len(e["row"])
        assert_topk_parity(lr[p, :m].astype(np.int64), ls[p, :m], e["row"], e["score"], what=f"lists row {i}")
    cfg = hip.SimilarityConfig(auto_link_threshold=0.6, dedup_threshold=0.95, contradiction_threshold=0.8)
    # similarity rule only: the generic walk == the fused GPU pass
    walk = linker.autolink_walk(h, scan_ids, [linker.similarity_rule(cfg)], max_edges_per_node=7)
    fr, to, w = h.autolink_pass_rows(scan, 100, cfg.auto_link_threshold, 7)
    assert [(a, b) for a, b, _, _ in walk] == list(zip(fr.tolist(), to.tolist()))
    assert np.allclose([x[3] for x in walk], w, atol=SCORE_TOL)
    # plus a structural rule that fires for every neighbour: the cap is hit early and may be overshot by one
    def same_agent(node, nb, score):
        return [("same_agent", 0.5)]
    walk2 = linker.autolink_walk(h, scan_ids, [linker.similarity_rule(cfg), same_agent], max_edges_per_node=7)
    thr = np.float32(cfg.auto_link_threshold)
    exp = []
    for i in scan:
        e = o.search(rows[i], 100)
        cnt = 0
        for j, s in zip(e["row"], e["score"]):
            if int(j) == int(i):
                continue
            if np.float32(s) >= thr:
                exp.append((int(i), int(j), "related_to")); cnt += 1
            exp.append((int(i), int(j), "same_agent")); cnt += 1
            if cnt >= 7:
                break
    assert [(a, b, r) for a, b, r, _ in walk2] == exp


@pytest.mark.parametrize("n,d,n_scan", [(5003, 1024, 64), (2000, 768, 17), (1031, 1024, 1), (4096, 768, 64), (3001, 384, 64), (2500, 512, 33)])
def test_small_scan_sets_take_the_stream_filter(hip, oracle, n, d, n_scan):
    """Scan sets of <= 64 rows at dim 384 / 512 / 768 / 1024 (streaming ingest, config 5) run pair_filter_stream_kernel: ragged
    last tile (n not a multiple of 16), fewer scanned rows than a consumer wave holds, scanned rows anywhere."""
    rows = oracle.synth_rows(n, d)
    ids = ids_for(n)
    h = hip.HipIndex(d); h.insert_batch(ids, rows)
    o = oracle.OracleIndex(d); o.insert_batch(ids, rows)
    rng = np.random.default_rng(n + n_scan)
    scan = np.sort(rng.choice(n, size=n_scan, replace=False)).astype(np.uint32)
    scan[-1] = n - 1                      # a scanned row inside the ragged tail
    scan = np.unique(scan).astype(np.uint32)
    thr = np.float32(0.8)
    fr, to, w = h.autolink_pass_rows(scan, 100, float(thr), 50)
    want = o.autolink_pass(scan, 100, thr, 50)
    got = {}
    for a, b, s in zip(fr, to, w):
        got.setdefault(int(a), []).append((int(b), float(s)))
    exp = {}
    for e in want:
        exp.setdefault(int(e["from_row"]), []).append((int(e["to_row"]), float(e["weight"])))
    assert got.keys() == exp.keys()
    for node in exp:
        gs, es = dict(got[node]), dict(exp[node])
        for nb in set(gs) ^ set(es):      # only pairs sitting on the threshold may differ
            sc = gs.get(nb, es.get(nb))
            assert abs(sc - float(thr)) <= SCORE_TOL, f"node {node} neighbour {nb} score {sc}"
        for nb in set(gs) & set(es):
            assert abs(gs[nb] - es[nb]) <= SCORE_TOL


def test_symmetric_rescore_is_bit_identical(hip):
    """In the symmetric pass each pair is scored once and written into both lists (allpairs.hip: pair_score_kernel).
    score(i, j) == score(j, i) bit for bit, so the edges — ids, order and weights — must be the bytes the one-sided
    rescore produces.  The switch is read once per process, hence two child processes."""
    import hashlib
    import os
    import subprocess
    import sys
    code = r'''
import sys, hashlib, numpy as np
sys.path.insert(0, %r)
import torch, cortex_amd
from cortex_amd import _lib
L = _lib.load()
out = []
for n, d, cap_thr in ((30000, 768, 0.85), (12000, 384, 0.3)):
    gen = torch.empty((n, d), dtype=torch.float32, device="cuda:0")
    assert L.cx_synth_fill_dev(0, gen.data_ptr(), 20260313, 20260313, 20260315, n // 50, 0, n, d, 1) == 0
    ids = np.zeros((n, 16), np.uint8); ids[:, 8:] = np.arange(n, dtype=np.uint64).astype(">u8").view(np.uint8).reshape(n, 8)
    h = cortex_amd.HipIndex(d); h.insert_batch_dev(ids, gen.data_ptr(), n, d)
    for r in (5, 77, 1234): h.remove(ids[r].tobytes())
    fr, to, w = h.autolink_pass_rows(None, 100, cap_thr, 50)     # 0.3: candidate lists overflow, rows take the exact path
    m = hashlib.sha256(); m.update(fr.tobytes()); m.update(to.tobytes()); m.update(w.tobytes())
    out.append("%%d %%s" %% (len(fr), m.hexdigest()))
print("|".join(out))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    # candidate capacity 512 (default) and 24: with 24 most lists overflow and are redone on the exact path, so the
    # "partner's list will be redone, score it yourself" rule of the symmetric kernels is exercised too; the edges do
    # not depend on the capacity either
    for sym, cap in (("1", "512"), ("0", "512"), ("1", "24"), ("0", "24")):
        env = dict(os.environ, CX_RESCORE_SYMMETRIC=sym, CX_PAIR_CAND_CAP=cap)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(r.stdout.strip().splitlines()[-1])
    assert res[0] == res[1] == res[2] == res[3] and int(res[0].split()[0]) > 100000, res


@pytest.mark.parametrize("env", [
    {"CX_PAIR_P_GRID": "8"},                              # one block per XCD: each walks ~10-30 tiles, claimed with s_atomic_add
    {"CX_PAIR_P_GRID": "8", "CX_PAIR_P_DYN": "0"},        # the same, statically dealt
    {"CX_PAIR_P_GRID": "24"},
    {},                                                   # the launch the product uses (one block per CU)
])
def test_persistent_filter_kernel_equals_the_per_tile_kernel(hip, oracle, monkeypatch, env):
    """allpairs_p.hip: persistent blocks whose LDS ring runs through the tile boundaries and whose hits leave the GEMM as
    records -> pairs.  With few blocks every block walks many tiles (the boundary code), at threshold 0.3 a tile holds
    more hit lanes than records and more hits than the LDS list (in-place walk, flush, direct appends).  Both filters feed the same exact rescore, so the edges must be the
    same bytes as the per-tile kernel's (pair_filter_kernel, 128 x 128 tiles), and equal to the oracle's (auto_linker.rs:215-264)."""
    for n, d, thr in ((5000, 768, 0.85), (4000, 384, 0.3), (2100, 1024, 0.75)):
        rows = oracle.synth_rows(n, d)
        h, o, ids = build(hip, oracle, rows)
        thr32 = float(np.float32(thr))
        monkeypatch.setenv("CX_PAIR_PERSIST", "0")
        ref = h.autolink_pass_rows(None, 100, thr32, 50)
        monkeypatch.setenv("CX_PAIR_PERSIST", "1")
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = h.autolink_pass_rows(None, 100, thr32, 50)
        for k in env:
            monkeypatch.delenv(k)
        assert len(ref[0]) > 1000
        for a, b in zip(got, ref):
            assert np.array_equal(a, b), f"n={n} d={d} thr={thr} env={env}"
        if d == 768:
            e = o.autolink_pass(np.arange(n), 100, thr32, 50, n_threads=8)
            compare_edges(per_node(*got), per_node(e["from_row"], e["to_row"], e["weight"]), thr32, oracle_scores(o, rows),
                          f"persistent n={n}")
        # a run of consecutive rows (an ingest tick's batch) is a range of the tiled shadow too: starts that are not on a
        # 32-row boundary (the swizzle period of the tiled shadow), in an odd 16-row block, and a range that ends the store
        for lo_s, m_s in ((1003, 700), (n - 533, 533), (16 * 77, 300)):
            scan = np.arange(lo_s, lo_s + m_s, dtype=np.uint32)
            monkeypatch.setenv("CX_PAIR_PERSIST", "0")
            ref = h.autolink_pass_rows(scan, 100, thr32, 50)
            monkeypatch.setenv("CX_PAIR_PERSIST", "1")
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            got = h.autolink_pass_rows(scan, 100, thr32, 50)
            for k in env:
                monkeypatch.delenv(k)
            assert len(ref[0]) > 100
            for a, b in zip(got, ref):
                assert np.array_equal(a, b), f"range scan [{lo_s}, {lo_s + m_s}) n={n} d={d} thr={thr} env={env}"
        # round 4: a LIST of rows (a cycle's batch in arbitrary order, with a repeated row and rows from the store's last 16-row
        # block) is gathered into a staged I panel and runs through the same persistent kernel — same bytes as the per-tile kernel
        rng = np.random.default_rng(n)
        for m_s in (700, 257, 130):
            scan = rng.permutation(n)[:m_s].astype(np.uint32)
            scan[5] = scan[9]
            scan[0], scan[1] = n - 1, n - 3
            monkeypatch.setenv("CX_PAIR_PERSIST", "0")
            ref = h.autolink_pass_rows(scan, 100, thr32, 50)
            monkeypatch.setenv("CX_PAIR_PERSIST", "1")
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            got = h.autolink_pass_rows(scan, 100, thr32, 50)
            for k in env:
                monkeypatch.delenv(k)
            assert len(ref[0]) > 50
            for a, b in zip(got, ref):
                assert np.array_equal(a, b), f"list scan of {m_s} rows n={n} d={d} thr={thr} env={env}"
            if d == 768 and m_s == 700:
                e = o.autolink_pass(scan, 100, thr32, 50, n_threads=8)
                compare_edges(per_node(*got), per_node(e["from_row"], e["to_row"], e["weight"]), thr32, oracle_scores(o, rows), "list scan vs oracle")


def test_removed_rows_are_not_scanned(hip, oracle):
    """A row removed from the index has no embedding any more: as a scanned node it proposes nothing
    (auto_linker.rs:217-218 skips nodes without an embedding), whether every row is scanned or it is named in the
    scan list; as a neighbour it never shows up.  Found by scripts/fuzz_autolink.py."""
    n, d = 600, 128
    rows = oracle.synth_rows(n, d)
    h, o, ids = build(hip, oracle, rows)
    gone = [2, 21, 26, 599]
    for r in gone:
        h.remove(ids[r].tobytes()); o.remove(ids[r].tobytes())
    thr = float(np.float32(0.75))
    for scan in (None, np.array([1, 2, 3, 21, 300, 599], dtype=np.uint32)):
        fr, to, w = h.autolink_pass_rows(scan, 100, thr, 50)
        assert not set(fr.tolist()) & set(gone) and not set(to.tolist()) & set(gone)
        want = o.autolink_pass(np.arange(n, dtype=np.uint32) if scan is None else scan, 100, np.float32(thr), 50)
        compare_edges(per_node(fr, to, w), per_node(want["from_row"], want["to_row"], want["weight"]), thr, oracle_scores(o, rows), "removed")
    lr, ls, lc = h.topk_lists_rows(100, np.array([1, 2, 21, 300], dtype=np.uint32))
    assert lc[1] == 0 and lc[2] == 0 and lc[0] == 100 and lc[3] == 100 and not set(lr[0, :100].tolist()) & set(gone)
    pairs = h.dedup_scan_rows(0.9) if hasattr(h, "dedup_scan_rows") else None
    if pairs is not None:
        assert not (set(np.asarray(pairs[0]).tolist()) | set(np.asarray(pairs[1]).tolist())) & set(gone)


@pytest.mark.parametrize("n,d,topk", [(6000, 768, 100), (5000, 384, 100), (3000, 1024, 20)])
def test_topk_lists_of_many_rows_filter_path(hip, oracle, n, d, topk, monkeypatch):
    """cx_topk_lists_rows for >= 256 scanned rows runs the all-pairs machinery (bf16 filter GEMM at a sampled threshold
    + exact rescore; rows whose list comes out short or overflowed take the exact path).  The lists must be the exact
    ordered top-k all the same: against the single-query scan for a sample of rows, against the oracle for a few, and —
    with a candidate cap so small that most rows take the exact path — against themselves."""
    rows = oracle.synth_rows(n, d)
    h, o, ids = build(hip, oracle, rows)
    for r in (7, 1500):
        h.remove(ids[r].tobytes()); o.remove(ids[r].tobytes())
    rng = np.random.default_rng(9)
    row_of = lambda gi: gi[:, 8:].copy().view(">u8").reshape(-1).astype(np.int64)

    def check(scan, lr, ls, lc, sample):
        scan_o = np.arange(n, dtype=np.uint32) if scan is None else scan
        assert lr.shape == (len(scan_o), topk)
        for p in sample:
            node = int(scan_o[p])
            m = int(lc[p])
            if node in (7, 1500):
                assert m == 0                                  # a removed row has no embedding: no list
                continue
            assert m == topk
            gi, gs, _ = h.search_arrays(rows[node], topk)
            assert_topk_parity(lr[p, :m].astype(np.int64), ls[p, :m], row_of(gi), gs, what=f"lists row {node}")
            assert np.all(np.diff(ls[p, :m]) <= 0) and len(set(lr[p, :m].tolist())) == m
            assert not ({7, 1500} & set(lr[p, :m].tolist()))
        for p in sample[:4]:
            node = int(scan_o[p])
            if node in (7, 1500):
                continue
            e = o.search(rows[node], topk)
            assert_topk_parity(lr[p, :topk].astype(np.int64), ls[p, :topk], e["row"], e["score"], what=f"lists row {node} vs oracle")

    lr, ls, lc = h.topk_lists_rows(topk, None)                    # every row: the symmetric pass
    check(None, lr, ls, lc, [0, 7, 8, 1499, 1500, n - 1] + rng.choice(n, 60, replace=False).tolist())
    scan = rng.permutation(n)[:700].astype(np.uint32)             # a cycle's batch, arbitrary order
    scan[3] = 7
    lr2, ls2, lc2 = h.topk_lists_rows(topk, scan)
    check(scan, lr2, ls2, lc2, list(range(0, 700, 9)))
    # the same lists whichever path produced them: subset vs whole store
    for p in range(0, 700, 13):
        if int(scan[p]) != 7:
            assert np.array_equal(lr2[p], lr[int(scan[p])]) and np.array_equal(ls2[p], ls[int(scan[p])])
    # ... and with a candidate cap of 16 (most rows overflow -> exact path): identical lists
    monkeypatch.setenv("CX_PAIR_CAND_CAP", "16")
    lr3, ls3, lc3 = h.topk_lists_rows(topk, scan)
    assert np.array_equal(lc3, lc2) and np.array_equal(lr3, lr2) and np.array_equal(ls3, ls2)


def test_small_scan_sets_through_the_worker_service_pass():
    """On shards of >= 131,072 rows the filter of a small scan set runs batchs.hip's threshold mode (workers + service wave,
    tiles claimed dynamically) instead of pair_filter_stream_kernel.  The switch is read once per process: the small-scan-set
    cases above, the streaming-ingest ticks on a bf16 store and the sharded lists run again in a child process with the row
    minimum lowered to test sizes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(root, "tests", "test_hip_autolink.py"),
                        os.path.join(root, "tests", "test_hip_bf16_store.py"), "-k",
                        "small_scan_sets_take_the_stream_filter or streaming_ingest_tick or pass_sees_upserts"],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, CX_BATCHS_MIN_ROWS="256"), cwd=root)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
