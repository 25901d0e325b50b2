"""The HIP cross-shard merge kernel (cx_merge_topk_dev) and the sharded search on one GPU:
the corpus is split into P row-range shards (P indexes on the same device), each shard's local
top-k lands in its chunk of the "gathered" buffer exactly as the all-gather would place it, and the
merged result must equal (a) the reference merge, (b) a single index over the whole corpus, (c) the oracle."""
import numpy as np
import pytest
import torch

from conftest import SCORE_TOL, assert_topk_parity, ids_for

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,d,k,nq,parts", [
    (4000, 384, 10, 5, 4), (3000, 768, 10, 64, 8), (1000, 128, 100, 3, 3), (90, 64, 16, 2, 8),
])
def test_sharded_merge_equals_single_index(hip, oracle, n, d, k, nq, parts):
    from cortex_amd.sharded import ShardedKnn, hip_local_fn, reference_merge, _hip_merge
    dev = torch.device("cuda", 0)
    rows = oracle.synth_rows(n, d)
    qs = oracle.synth_queries(n, d, nq)
    ids = ids_for(n)
    cuts = np.linspace(0, n, parts + 1).astype(int)
    if parts >= 3:
        cuts[1] = cuts[0] + 3  # one shard smaller than k
    shards = []
    for p in range(parts):
        h = hip.HipIndex(d)
        h.insert_batch(ids[cuts[p]:cuts[p + 1]], rows[cuts[p]:cuts[p + 1]])
        shards.append(h)
    dq = torch.from_numpy(qs).to(dev)
    bases = cuts[:-1].tolist()
    # run every shard as "rank p" and place its chunk where the all-gather would
    driver = ShardedKnn(0, parts, bases, nq, k, dev, lambda *a: None)
    for p in range(parts):
        s = ShardedKnn(p, parts, bases, nq, k, dev, hip_local_fn(shards[p]))
        kk = min(k, len(shards[p]))
        if kk < k:  # a shard with fewer rows than k answers with what it has (per-shard k is clamped)
            s2 = ShardedKnn(p, parts, bases, nq, kk, dev, hip_local_fn(shards[p]))
            s2.local_fn(dq.data_ptr(), nq, s2)
            r2, sc2, di2, c2 = s2.chunk_views(s2.local)
            r, sc, di, c = s.chunk_views(s.local)
            r[:, :kk], sc[:, :kk], di[:, :kk] = r2, sc2, di2
            c.copy_(c2)
        else:
            s.local_fn(dq.data_ptr(), nq, s)
        driver.gathered[p * driver.words:(p + 1) * driver.words].copy_(s.local)
    torch.cuda.synchronize()
    _hip_merge(driver)
    torch.cuda.synchronize()
    got = (driver.out_rows.cpu().numpy().copy(), driver.out_scores.cpu().numpy().copy(),
           driver.out_dists.cpu().numpy().copy(), driver.out_counts.cpu().numpy().copy())
    reference_merge(driver)
    torch.cuda.synchronize()
    for a, b in zip(got, (driver.out_rows.cpu().numpy(), driver.out_scores.cpu().numpy(),
                          driver.out_dists.cpu().numpy(), driver.out_counts.cpu().numpy())):
        cnt = got[3]
        if a.ndim == 2:
            for qi in range(nq):
                assert np.array_equal(a[qi, :cnt[qi]], b[qi, :cnt[qi]], equal_nan=True), "HIP merge != reference merge"
        else:
            assert np.array_equal(a, b)
    o = oracle.OracleIndex(d)
    o.insert_batch(ids, rows)
    for qi in range(nq):
        e = o.search(qs[qi], k)
        m = int(got[3][qi])
        assert_topk_parity(got[0][qi, :m], got[1][qi, :m], e["row"], e["score"], what=f"sharded q{qi}")


def test_world_one_is_the_local_list(hip, oracle):
    from cortex_amd.sharded import ShardedKnn, hip_local_fn
    dev = torch.device("cuda", 0)
    rows = oracle.synth_rows(2000, 768)
    qs = oracle.synth_queries(2000, 768, 2)
    ids = ids_for(2000)
    h = hip.HipIndex(768)
    h.insert_batch(ids, rows)
    s = ShardedKnn(0, 1, [0], 2, 10, dev, hip_local_fn(h))
    dq = torch.from_numpy(qs).to(dev)
    s.search(dq.data_ptr())
    torch.cuda.synchronize()
    r, sc, di, c = s.chunk_views(s.local)
    o = oracle.OracleIndex(768)
    o.insert_batch(ids, rows)
    for qi in range(2):
        e = o.search(qs[qi], 10)
        assert int(c[qi]) == 10
        assert_topk_parity(r[qi].cpu().numpy(), sc[qi].cpu().numpy(), e["row"], e["score"], what="world=1")


def test_submit_rotates_batches_over_streams(hip, oracle):
    """submit() on a GPU enqueues a stream of batches on several HIP streams in rotation, each with its own exchange and result
    buffers; after flush() the caller's stream has every batch: seven different batches (more than there are streams, so
    buffers are reused), each read back right after its own flush, equal the oracle's answers."""
    from cortex_amd.sharded import ShardedKnn, hip_local_fn
    dev = torch.device("cuda", 0)
    n, d, nq, k = 5000, 768, 8, 10
    rows = oracle.synth_rows(n, d)
    ids = ids_for(n)
    h = hip.HipIndex(d)
    h.insert_batch(ids, rows)
    o = oracle.OracleIndex(d)
    o.insert_batch(ids, rows)
    s = ShardedKnn(0, 1, [0], nq, k, dev, hip_local_fn(h))
    assert s._streams is not None and len(s._streams) >= 2
    all_q = oracle.synth_queries(n, d, 7 * nq)
    dq = torch.from_numpy(all_q).to(dev)
    views = []
    for b in range(7):
        s.submit(dq.data_ptr() + b * nq * d * 4)
        views.append(s.chunk_views(s.local))     # world == 1: the local list of the slot this batch used
        if b % 3 == 2:                            # flush now and then: the batches so far must be complete
            s.flush()
            torch.cuda.current_stream().synchronize()
            for bb in range(b - 2, b + 1):
                r, sc, di, c = views[bb]
                for qi in range(nq):
                    e = o.search(all_q[bb * nq + qi], k)
                    assert int(c[qi]) == k
                    assert_topk_parity(r[qi].cpu().numpy(), sc[qi].cpu().numpy(), e["row"], e["score"], what=f"batch {bb} q{qi}")
    s.flush()
    torch.cuda.synchronize()
    r, sc, di, c = views[6]
    for qi in range(nq):
        e = o.search(all_q[6 * nq + qi], k)
        assert_topk_parity(r[qi].cpu().numpy(), sc[qi].cpu().numpy(), e["row"], e["score"], what=f"last batch q{qi}")


@pytest.mark.parametrize("n,d,parts,thr,block", [(3000, 768, 3, 0.85, 1024), (2000, 384, 4, 0.75, 512), (900, 100, 2, 0.8, 400)])
def test_sharded_autolink_equals_single_index_pass(hip, oracle, n, d, parts, thr, block):
    """The all-pairs pass over row-range shards (external-query lists + all-gather layout + merge + rule walk)
    proposes exactly the edges of the single-index pass; P shards live on the one GPU, the all-gather is
    emulated by placing every shard's chunk where the collective would."""
    from cortex_amd.sharded import ShardedAutolink, ShardedKnn, hip_lists_fn, hip_rows_fn, _hip_merge
    dev = torch.device("cuda", 0)
    rows = oracle.synth_rows(n, d)
    ids = ids_for(n)
    cuts = np.linspace(0, n, parts + 1).astype(int)
    shards = []
    for p in range(parts):
        h = hip.HipIndex(d)
        h.insert_batch(ids[cuts[p]:cuts[p + 1]], rows[cuts[p]:cuts[p + 1]])
        shards.append(h)
    whole = hip.HipIndex(d)
    whole.insert_batch(ids, rows)
    thr32 = float(np.float32(thr))
    rng = np.random.default_rng(5)
    deleted = (rng.random(n) < 0.05)
    fr, to, w = whole.autolink_pass_rows(None, 100, thr32, 50, deleted.astype(np.uint8))
    want = {}
    for a, b, s in zip(fr, to, w):
        want.setdefault(int(a), []).append((int(b), float(s)))

    sizes = [int(cuts[p + 1] - cuts[p]) for p in range(parts)]
    got = {}
    # emulate the ranks: for every block, every "rank" computes its lists; rank `src` merges and walks the rules
    drivers = [ShardedAutolink(p, parts, sizes, d, 100, dev, hip_lists_fn(shards[p], thr32), hip_rows_fn(shards[p]),
                               block=block) for p in range(parts)]
    bases = drivers[0].bases
    for src in range(parts):
        for lo in range(0, sizes[src], block):
            m = min(block, sizes[src] - lo)
            buf = drivers[src].buf
            buf.zero_()
            drivers[src].rows_fn(lo, m, buf)
            owner = drivers[src].knn
            for p in range(parts):
                k = drivers[p].knn
                k.local_fn(buf, block, k)
                torch.cuda.synchronize()
                owner.gathered[p * owner.words:(p + 1) * owner.words].copy_(k.local)
            _hip_merge(owner)
            torch.cuda.synchronize()
            rws = owner.out_rows[:m].cpu().numpy(); sc = owner.out_scores[:m].cpu().numpy(); cnt = owner.out_counts[:m].cpu().numpy()
            self_g = bases[src] + lo + np.arange(m)
            valid = np.arange(100)[None, :] < cnt[:, None]
            ok = valid & (rws != self_g[:, None]) & (sc >= np.float32(thr32)) & ~deleted[np.where(valid, rws, 0)]
            ok &= np.cumsum(ok, axis=1) <= 50
            for i, j in zip(*np.nonzero(ok)):
                got.setdefault(int(self_g[i]), []).append((int(rws[i, j]), float(sc[i, j])))
    assert got.keys() == want.keys()
    for node in want:
        assert [x[0] for x in got[node]] == [x[0] for x in want[node]], f"node {node}"
        assert np.allclose([x[1] for x in got[node]], [x[1] for x in want[node]], atol=0, rtol=0)


def test_sharded_autolink_world_one_run(hip, oracle):
    from cortex_amd.sharded import ShardedAutolink, hip_alive_fn, hip_lists_fn, hip_rows_fn
    dev = torch.device("cuda", 0)
    n, d = 1500, 768
    rows = oracle.synth_rows(n, d)
    h = hip.HipIndex(d)
    ids = ids_for(n)
    h.insert_batch(ids, rows)
    for r in (3, 700, 1499):             # removed rows are neither neighbours nor scanned (auto_linker.rs:217-218)
        h.remove(ids[r].tobytes())
    assert not h.rows_alive(3, 1)[0] and h.rows_alive(4, 1)[0] and int(h.rows_alive().sum()) == n - 3
    thr32 = float(np.float32(0.85))
    sa = ShardedAutolink(0, 1, [n], d, 100, dev, hip_lists_fn(h, thr32), hip_rows_fn(h), block=512, alive_fn=hip_alive_fn(h))
    f, t, w = sa.run(thr32, 50)
    assert not set(f.tolist()) & {3, 700, 1499}
    fr, to, ww = h.autolink_pass_rows(None, 100, thr32, 50)
    assert np.array_equal(f, fr.astype(np.int64)) and np.array_equal(t, to.astype(np.int64)) and np.array_equal(w, ww)
