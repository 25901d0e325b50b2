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
