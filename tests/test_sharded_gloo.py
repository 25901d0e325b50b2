"""N > 1 path on CPU: world_size-2 gloo run of cortex_amd.sharded.ShardedKnn.

The row-range sharding, the packed all-gather layout, the shard row bases and the merge order are
product code (cortex_amd/sharded.py); the local searcher is injected — here the CPU oracle, on the GPU
the HIP index — so the collective plumbing is exercised without a GPU.  The HIP merge kernel itself is
checked against the same reference merge in tests/test_hip_sharded.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank: int, world: int, port: int, n_total: int, d: int, k: int, nq: int, ragged: bool, pipelined: bool, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from cortex_amd.sharded import ShardedKnn, reference_merge
        from conftest import ids_for
        # ragged: shard sizes differ and one shard is smaller than k
        if ragged:
            sizes = [n_total - 3, 3][:world] if world == 2 else [n_total // world] * world
        else:
            sizes = [n_total // world] * world
        bases = np.concatenate([[0], np.cumsum(sizes)[:-1]]).tolist()
        rows = O.synth_rows(n_total, d)
        qs = O.synth_queries(n_total, d, nq)
        lo, hi = bases[rank], bases[rank] + sizes[rank]
        local = O.OracleIndex(d)
        local.insert_batch(ids_for(n_total)[lo:hi], rows[lo:hi])

        def local_fn(queries, nq_, s):
            r, sc, di, cnt = s.chunk_views(s.local)
            for qi in range(nq_):
                e = local.search(queries[qi], s.k)
                m = len(e)
                cnt[qi] = m
                r[qi, :m] = torch.from_numpy(e["row"].astype(np.int32))
                sc[qi, :m] = torch.from_numpy(e["score"].copy())
                di[qi, :m] = torch.from_numpy(e["distance"].copy())

        knn = ShardedKnn(rank, world, bases, nq, k, torch.device("cpu"), local_fn, merge_fn=reference_merge)
        if pipelined:
            # stream of three batches; the answer checked below is the last one's
            knn.submit(O.synth_queries(n_total, d, nq) * np.float32(-1.0))
            knn.submit(qs[::-1].copy())
            knn.submit(qs)
            knn.flush()
        else:
            knn.search(qs)
        # every rank holds the same merged answer; compare with one oracle over the whole corpus
        full = O.OracleIndex(d)
        full.insert_batch(ids_for(n_total), rows)
        ok = True
        for qi in range(nq):
            e = full.search(qs[qi], k)
            m = int(knn.out_counts[qi])
            ok &= m == len(e)
            ok &= np.array_equal(knn.out_rows[qi, :m].numpy(), e["row"].astype(np.int64))
            ok &= np.array_equal(knn.out_scores[qi, :m].numpy(), e["score"])
            ok &= np.array_equal(knn.out_dists[qi, :m].numpy(), e["distance"])
        out_q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total,d,k,nq,ragged,pipelined", [(400, 64, 10, 3, False, False), (120, 32, 7, 2, True, False),
                                                             (300, 64, 10, 2, False, True)])
def test_two_rank_gloo_sharded_search_equals_single_index(n_total, d, k, nq, ragged, pipelined):
    from oracle import oracle as O
    O.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, d, k, nq, ragged, pipelined, q)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got == [(0, True), (1, True)]


def test_packed_layout():
    from cortex_amd.sharded import packed_words
    assert packed_words(1, 10) == 32 and packed_words(64, 10) % 4 == 0 and packed_words(64, 10) >= 3 * 640 + 64


def _autolink_worker(rank: int, world: int, port: int, n_total: int, d: int, thr: float, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from cortex_amd.sharded import ShardedAutolink, reference_merge
        from conftest import ids_for
        sizes = [n_total - n_total // 3, n_total // 3]
        bases = [0, sizes[0]]
        rows = O.synth_rows(n_total, d)
        lo, hi = bases[rank], bases[rank] + sizes[rank]
        local = O.OracleIndex(d)
        local.insert_batch(ids_for(n_total)[lo:hi], rows[lo:hi])
        thr32 = float(np.float32(thr))
        topk = 100

        def lists_fn(queries, nq, s):  # the oracle stands in for cx_autolink_lists_dev
            r, sc, di, cnt = s.chunk_views(s.local)
            q = queries.numpy()
            for qi in range(getattr(s, "live", nq)):
                e = local.search_threshold(q[qi], thr32)[:topk]
                m = len(e)
                cnt[qi] = m
                r[qi, :m] = torch.from_numpy(e["row"].astype(np.int32))
                sc[qi, :m] = torch.from_numpy(e["score"].copy())
                di[qi, :m] = torch.from_numpy(e["distance"].copy())
            cnt[getattr(s, "live", nq):] = 0

        def rows_fn(row_lo, n, buf):
            buf[:n] = torch.from_numpy(rows[lo + row_lo:lo + row_lo + n])

        sa = ShardedAutolink(rank, world, sizes, d, topk, torch.device("cpu"), lists_fn, rows_fn,
                             merge_fn=reference_merge, block=64)
        deleted = (np.arange(n_total) % 17 == 3)
        f, t, w = sa.run(thr32, 50, deleted)
        full = O.OracleIndex(d)
        full.insert_batch(ids_for(n_total), rows)
        e = full.autolink_pass(np.arange(lo, hi), topk, thr32, 50, deleted.astype(np.uint8))
        ok = (np.array_equal(f, e["from_row"].astype(np.int64)) and np.array_equal(t, e["to_row"].astype(np.int64))
              and np.array_equal(w, e["weight"]))
        # the rescan (auto_linker.rs:137-182): ~30 % of the own rows already carry related_to edges (dropped WITHOUT counting,
        # :249-258), a tight per-node cap, cap 0 (the reference still lets the first neighbour through, :259-262) and the
        # per-cycle cap over BOTH ranks' proposals in scan order (:284-287)
        rng = np.random.default_rng(5)
        have = {}
        for a, b in zip(e["from_row"], e["to_row"]):
            have.setdefault(int(a), []).append(int(b))
        lists = [sorted(rng.permutation(have.get(r, []))[:3].tolist()) + [int(rng.integers(0, n_total))] if rng.random() < 0.3 else []
                 for r in range(lo, hi)]
        eo = np.zeros(len(lists) + 1, np.uint64); eo[1:] = np.cumsum([len(x) for x in lists])
        et = np.array([x for l in lists for x in l], dtype=np.uint32)
        whole = full.autolink_pass(np.arange(n_total), topk, thr32, 4, deleted.astype(np.uint8))   # both ranks' rows, no edges yet: sizes the cycle cap
        for cap_node, cyc in ((4, None), (0, None), (4, len(whole) // 2)):
            f2, t2, w2 = sa.run(thr32, cap_node, deleted, existing=(eo, et), max_edges_per_cycle=cyc)
            # the oracle walks ALL scanned rows of the cycle in order with every node's own existing set: rank 1's rows carry theirs
            # only on rank 1, so each rank checks its own rows with the other rank's rows edge-free — as the other rank does
            other = np.arange(0, lo) if rank == 1 else np.arange(hi, n_total)
            if rank == 1:
                scan_all = np.concatenate([other, np.arange(lo, hi)]); eo_all = np.concatenate([np.zeros(len(other), np.uint64), eo])
            else:
                scan_all = np.concatenate([np.arange(lo, hi), other]); eo_all = np.concatenate([eo, np.full(len(other), eo[-1], np.uint64)])
            ref = full.autolink_pass(scan_all, topk, thr32, cap_node, deleted.astype(np.uint8), existing=(eo_all, et),
                                     max_edges_per_cycle=None)
            mine_ref = ref[(ref["from_row"] >= lo) & (ref["from_row"] < hi)]
            if cyc is not None:   # budget left after the ranks in front (rank 0's proposals count first)
                r0 = full.autolink_pass(np.arange(0, sizes[0]), topk, thr32, cap_node, deleted.astype(np.uint8),
                                        existing=(eo, et) if rank == 0 else None)
                before = 0 if rank == 0 else None
                if rank == 1:
                    before = -1   # rank 0 reports its own count through the collective; checked below by the total
                keep = cyc if rank == 0 else None
                if rank == 0:
                    mine_ref = mine_ref[:max(0, min(len(mine_ref), cyc))]
                else:
                    mine_ref = mine_ref[:len(f2)]   # what rank 0 left of the budget: the total is checked by the parent
            ok = ok and np.array_equal(f2, mine_ref["from_row"].astype(np.int64)) and np.array_equal(t2, mine_ref["to_row"].astype(np.int64)) \
                and np.array_equal(w2, mine_ref["weight"])
            if cap_node == 0:
                ok = ok and len(f2) > 0 and np.all(np.bincount((f2 - lo).astype(np.int64), minlength=hi - lo) <= 1)
            out_q.put(("cycle", rank, cap_node, cyc, len(f2)))
        out_q.put((rank, bool(ok), len(f)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharded_autolink_equals_oracle_pass():
    from oracle import oracle as O
    O.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_autolink_worker, args=(r, 2, port, 300, 64, 0.75, q)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(180) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    msgs = []
    while True:
        try:
            msgs.append(q.get(timeout=5))
        except Exception:
            break
    got = sorted(m for m in msgs if m[0] != "cycle")
    assert [g[:2] for g in got] == [(0, True), (1, True)] and all(g[2] > 0 for g in got)
    # the per-cycle cap holds over both ranks together, and rank 0's proposals come first
    capped = {m[1]: m[4] for m in msgs if m[0] == "cycle" and m[3] is not None}
    budget = [m[3] for m in msgs if m[0] == "cycle" and m[3] is not None][0]
    assert capped[0] + capped[1] == budget and capped[0] > 0
