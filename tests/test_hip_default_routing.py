"""The screening pass (batchs.hip) at sizes the DEFAULT routing sends to it — no CX_BATCHS_MIN_ROWS — against the oracle
and the single-query scan:

  * several passes in flight at once (its blocks wait for each other across the grid, batchs.hip: warm-up and hit rings):
    `ShardedKnn.submit` rotating 64-query batches over its HIP streams, and four host threads in `search_batch` at the
    same time (the trait's `&self` contract, vector/index.rs:390-410 under `RwLock::read`), on a 200k-row store;
  * its threshold mode (the filter of a small scan set: BASELINE configs[4]'s 64-row ingest tick, and an arbitrary small
    scan set) on a bf16 store of >= 131,072 rows."""
import threading

import numpy as np
import pytest
import torch

from conftest import assert_topk_parity, ids_for

pytestmark = pytest.mark.gpu


def _row_of(id16) -> int:
    return int.from_bytes(bytes(id16[8:]), "big")


def test_concurrent_screening_passes_at_default_routing(hip, oracle):
    from cortex_amd import _lib
    from cortex_amd.sharded import ShardedKnn, hip_local_fn
    L = _lib.load()
    dev = torch.device("cuda", 0)
    n, d, k, nq, nb = 200_000, 384, 10, 64, 9
    gen = torch.empty((n, d), dtype=torch.float32, device=dev)
    assert L.cx_synth_fill_dev(0, gen.data_ptr(), 20260313, 20260313, 20260315, n // 50, 0, n, d, 1) == 0
    ids = np.zeros((n, 16), np.uint8)
    ids[:, 8:] = np.arange(n, dtype=np.uint64).astype(">u8").view(np.uint8).reshape(n, 8)
    h = hip.HipIndex(d)
    h.insert_batch_dev(ids, gen.data_ptr(), n, d)
    qs_t = torch.empty((nb * nq, d), dtype=torch.float32, device=dev)
    assert L.cx_synth_fill_dev(0, qs_t.data_ptr(), 20260313, 20260314, 20260315, n // 50, 0, nb * nq, d, 0) == 0
    qs = qs_t.cpu().numpy()
    # what every path must return: the single-query scan, itself pinned to the oracle on a sample (and in test_hip_parity.py)
    want = []
    for q in qs:
        gi, gs, gd = h.search_arrays(q, k)
        want.append((np.array([_row_of(x) for x in gi]), gs))
    o = oracle.OracleIndex(d)
    o.insert_batch(ids, gen.cpu().numpy())
    del gen
    for i in range(0, nb * nq, 24):
        e = o.search(qs[i], k)
        assert_topk_parity(want[i][0], want[i][1], e["row"], e["score"], what=f"scan vs oracle q{i}")

    # (1) a stream of batches on rotating HIP streams: more batches than streams, read back after one flush
    s = ShardedKnn(0, 1, [0], nq, k, dev, hip_local_fn(h))
    assert s._streams is not None and len(s._streams) >= 2
    views = []
    for b in range(nb):
        s.submit(qs_t.data_ptr() + b * nq * d * 4)
        views.append(s.chunk_views(s.local))     # world == 1: the local list of the slot this batch used
        if b % len(s._streams) == len(s._streams) - 1 or b == nb - 1:   # before a slot's buffers are reused: flush and check
            s.flush()
            torch.cuda.synchronize()
            for bb in range(b - (b % len(s._streams)), b + 1):
                r, sc, di, c = views[bb]
                for qi in range(nq):
                    assert int(c[qi]) == k
                    w = want[bb * nq + qi]
                    assert_topk_parity(r[qi].cpu().numpy(), sc[qi].cpu().numpy(), w[0], w[1], what=f"submit batch {bb} q{qi}")

    # (1b) the same with 128-query batches (two banks per pass at this row width): the shard asks for two streams, not four
    assert (h.search_batch_streams_hint(1), h.search_batch_streams_hint(64), h.search_batch_streams_hint(128)) == (1, 4, 2)
    s2 = ShardedKnn(0, 1, [0], 128, k, dev, hip_local_fn(h))
    assert s2._streams is not None and len(s2._streams) == 2
    nb2 = nb * nq // 128
    views = []
    for b in range(nb2):
        s2.submit(qs_t.data_ptr() + b * 128 * d * 4)
        views.append(s2.chunk_views(s2.local))
        if b % 2 == 1 or b == nb2 - 1:
            s2.flush()
            torch.cuda.synchronize()
            for bb in range(b - (b % 2), b + 1):
                r, sc, di, c = views[bb]
                for qi in range(128):
                    assert int(c[qi]) == k
                    w = want[bb * 128 + qi]
                    assert_topk_parity(r[qi].cpu().numpy(), sc[qi].cpu().numpy(), w[0], w[1], what=f"submit 128-query batch {bb} q{qi}")

    # (2) four host threads in search_batch at once, three rounds each
    errs = []

    def reader(t):
        try:
            for rnd in range(3):
                b = (t * 3 + rnd) % nb
                bi, bs, bd, bc = h.search_batch_arrays(qs[b * nq:(b + 1) * nq], k)
                for qi in range(nq):
                    assert int(bc[qi]) == k
                    w = want[b * nq + qi]
                    assert_topk_parity(np.array([_row_of(x) for x in bi[qi]]), bs[qi], w[0], w[1], what=f"thread {t} batch {b} q{qi}")
        except Exception as ex:   # noqa: BLE001
            errs.append(repr(ex))

    th = [threading.Thread(target=reader, args=(t,)) for t in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs[:3]


def _bf16_round(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(x.shape)


def test_small_scan_sets_on_a_large_bf16_store_at_default_routing(hip, oracle):
    """140k x 512 bf16 rows: the 64-row ingest tick (appended rows linked against the whole store), then an arbitrary scan set of
    40 old rows, then a second tick behind an in-place upsert — all through batchs_kernel<512, true>, which the default
    routing picks from 131,072 rows; edges must be the oracle's for the rounded vectors."""
    from test_hip_autolink import compare_edges, oracle_scores, per_node
    n0, d, batch = 140_000, 512, 64
    allrows = oracle.synth_rows(n0 + 2 * batch, d)
    ids = ids_for(n0 + 2 * batch)
    h = hip.HipIndex(d, dtype="bf16")
    h.reserve(n0 + 2 * batch)
    h.insert_batch(ids[:n0], allrows[:n0])
    o = oracle.OracleIndex(d)
    o.insert_batch(ids[:n0], _bf16_round(allrows[:n0]))
    thr = float(np.float32(0.85))
    lo = n0
    rng = np.random.default_rng(3)
    for tick in range(2):
        new = allrows[lo:lo + batch]
        dev = torch.from_numpy(new).to("cuda:0")
        h.insert_batch_dev(ids[lo:lo + batch], dev.data_ptr(), batch, d)
        o.insert_batch(ids[lo:lo + batch], _bf16_round(new))
        for scan, name in ((np.arange(lo, lo + batch, dtype=np.uint32), "tick"),
                           (rng.choice(lo + batch, 40, replace=False).astype(np.uint32), "arbitrary rows")):
            fr, to, w = h.autolink_pass_rows(scan, 100, thr, 50)
            e = o.autolink_pass(scan, 100, thr, 50, n_threads=8)
            got, exp = per_node(fr, to, w), per_node(e["from_row"], e["to_row"], e["weight"])
            assert len(exp) > 0 and set(got) <= set(int(x) for x in scan)
            compare_edges(got, exp, thr, oracle_scores(o, _bf16_round(allrows[:lo + batch])), f"{name}, tick {tick}")
        lo += batch
        if tick == 0:
            allrows[7] = allrows[lo + 3] * np.float32(1.5)
            h.insert(ids[7].tobytes(), allrows[7])
            o.insert(ids[7].tobytes(), _bf16_round(allrows[7:8])[0])
    # and the batched search of the same store (top-k mode of the same kernel, bf16 rows re-scored)
    qs = oracle.synth_queries(n0, d, 70)
    bi, bs, bd, bc = h.search_batch_arrays(qs, 10)
    lut = {ids[i].tobytes(): i for i in range(lo)}
    for i in range(0, 70, 5):
        e = o.search(qs[i], 10)
        m = int(bc[i])
        assert m == len(e["row"])
        assert_topk_parity(np.array([lut[g.tobytes()] for g in bi[i, :m]]), bs[i, :m], e["row"], e["score"], what=f"batch q{i}")
