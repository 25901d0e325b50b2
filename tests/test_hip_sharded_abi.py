"""The multi-GPU index behind the C ABI (cx_sharded_*): one process, one shard per listed device.
A one-GPU box lists device 0 several times (P shards on one GPU — same code path: per-shard streams, publish into
the root's gather buffer, merge_parts_kernel, host hand-over).

Bar: a sharded index is indistinguishable from a SINGLE HipIndex that saw the same sequence of calls — same ids in the
same order (ties: global insertion order), scores equal, and the row-indexed linker passes return the same edges in
global rows — and therefore matches the oracle wherever the single index does."""
import threading
import uuid

import numpy as np
import pytest

from conftest import SCORE_TOL, assert_topk_parity, ids_for

pytestmark = pytest.mark.gpu


def build(hip, rows, ids, shards=3):
    d = rows.shape[1]
    one = hip.HipIndex(d)
    sh = hip.ShardedHipIndex(d, [0] * shards)
    # several calls of different sizes: runs are cut at the 4096-row placement blocks
    cuts = [0, 1, 5000, 5003, len(rows)]
    for a, b in zip(cuts[:-1], cuts[1:]):
        one.insert_batch(ids[a:b], rows[a:b])
        sh.insert_batch(ids[a:b], rows[a:b])
    return one, sh


def same_lists(a, b, what):
    ai, asc, ad = a
    bi, bsc, bd = b
    assert len(asc) == len(bsc), f"{what}: {len(asc)} vs {len(bsc)} results"
    assert np.array_equal(ai, bi), f"{what}: ids differ"
    assert np.allclose(asc, bsc, rtol=0, atol=1e-6) and np.allclose(ad, bd, rtol=0, atol=1e-6), f"{what}: scores differ"


@pytest.fixture(scope="module")
def corpus(oracle):
    n, d = 13_000, 384
    rows = oracle.synth_rows(n, d)
    # six exact duplicates of one vector spread over the placement blocks (= over the shards): ties by insertion order
    for r in (10, 4200, 4300, 8500, 9000, 12999):
        rows[r] = rows[10]
    return rows, ids_for(n), oracle.synth_queries(n, d, 70)


def test_sharded_equals_single_index_and_oracle(hip, oracle, corpus):
    rows, ids, qs = corpus
    n, d = rows.shape
    one, sh = build(hip, rows, ids)
    assert sh.n_shards == 3 and len(sh) == len(one) == n and sh.row_count() == n
    assert [sh.shard_len(i) for i in range(3)] == [4096 + 712, 4096, 4096]      # placement blocks of 4096 new ids, round robin
    o = oracle.OracleIndex(d)
    o.insert_batch(ids, rows)
    # single query, k = 10 and the linker's 100; the duplicate vector as a query: six ties in insertion order
    for k in (1, 10, 100):
        for q in list(qs[:6]) + [rows[10]]:
            same_lists(sh.search_arrays(q, k), one.search_arrays(q, k), f"search k={k}")
    gi, gs, _ = sh.search_arrays(rows[10], 6)
    assert [int(x) for x in gi[:, 8:].copy().view(">u8").reshape(-1)] == [10, 4200, 4300, 8500, 9000, 12999]
    e = o.search(qs[0], 10)
    gi, gs, _ = sh.search_arrays(qs[0], 10)
    assert_topk_parity(gi[:, 8:].copy().view(">u8").reshape(-1).astype(np.int64), gs, e["row"], e["score"], what="sharded vs oracle")
    # batches: the MFMA kernel per shard (k = 10, wide lists k = 100), a ragged second pass
    for k in (10, 100):
        a = sh.search_batch_arrays(qs, k)
        b = one.search_batch_arrays(qs, k)
        assert np.array_equal(a[3], b[3])
        for j in range(len(qs)):
            c = int(a[3][j])
            same_lists((a[0][j, :c], a[1][j, :c], a[2][j, :c]), (b[0][j, :c], b[1][j, :c], b[2][j, :c]), f"batch k={k} q{j}")
    # k beyond the in-register lists (host merge of the shards' sort paths), and k beyond the corpus
    same_lists(sh.search_arrays(qs[1], 700), one.search_arrays(qs[1], 700), "k=700")
    assert len(sh.search_arrays(qs[1], 50_000)[1]) == n
    # threshold search: variable length, merged on the host
    for thr in (0.9, 0.6):
        same_lists(sh.search_threshold_arrays(qs[2], thr), one.search_threshold_arrays(qs[2], thr), f"threshold {thr}")
    # a query longer than the dimension (the reference zips, index.rs:172): dot over the prefix, |q| over all of it
    ql = np.concatenate([qs[3], np.float32([0.5, -0.25])])
    same_lists(sh.search_arrays(ql, 10), one.search_arrays(ql, 10), "long query")


def test_sharded_mutations_filters_and_rebuild(hip, oracle, corpus):
    rows, ids, qs = corpus
    n, d = rows.shape
    one, sh = build(hip, rows, ids)
    new = oracle.synth_queries(n, d, 4)
    extra_ids = ids_for(n + 300)[n:]
    extra = oracle.synth_rows(n + 300, d, n, 300)
    for ix in (one, sh):
        for i, r in enumerate((3, 4100, 9000, 12000)):               # in-place upserts keep shard and row
            ix.insert(ids[r].tobytes(), new[i])
        for r in (7, 4097, 8200, 8201, 12998):
            ix.remove(ids[r].tobytes())
        ix.remove(uuid.uuid4().bytes)                                 # unknown id: not an error
        ix.set_metadata(extra_ids[5].tobytes(), "decision", "kai")    # metadata BEFORE the vector (vector/tests.rs:65-66)
        ix.insert_batch(extra_ids, extra)
        for r in range(0, 3000, 2):
            ix.set_metadata(ids[r].tobytes(), ("fact", "decision", "event")[r % 3], ("kai", "test")[(r // 2) % 2])
        ix.insert(ids[7].tobytes(), rows[7])                          # a removed id comes back as a NEW row at the end
    assert len(sh) == len(one) and sh.row_count() == one.row_count()
    with pytest.raises(hip.ValidationError):
        sh.insert(uuid.uuid4().bytes, np.zeros(d + 1, np.float32))
    top = one.search_arrays(qs[0], 5)[0]
    filters = [None, hip.VectorFilter(kinds=["decision"]), hip.VectorFilter(kinds=["fact", "event"], source_agent="kai"),
               hip.VectorFilter(exclude=[top[0].tobytes(), top[2].tobytes(), uuid.uuid4().bytes]),
               hip.VectorFilter(kinds=["never-seen"], source_agent="nobody")]
    for f in filters:
        for k in (10, 100):
            for q in (qs[0], qs[5], extra[5], rows[7]):
                same_lists(sh.search_arrays(q, k, f), one.search_arrays(q, k, f), f"filter {f} k={k}")
        a, b = sh.search_batch_arrays(qs[:40], 10, f), one.search_batch_arrays(qs[:40], 10, f)
        assert np.array_equal(a[3], b[3]) and np.array_equal(a[0], b[0])
    # global rows are the single index's rows
    probe = [ids[0].tobytes(), ids[4096].tobytes(), extra_ids[299].tobytes(), ids[7].tobytes(), uuid.uuid4().bytes]
    assert np.array_equal(sh.rows_of(probe), one.rows_of(probe))
    for r in (0, 4096, 9999, one.row_count() - 1):
        assert sh.row_id(r) == one.row_id(r)
    # rebuild compacts every shard and renumbers the global rows in order — like the single index's compaction
    one.rebuild(); sh.rebuild()
    assert sh.row_count() == one.row_count() == len(one)
    assert np.array_equal(sh.rows_of(probe), one.rows_of(probe))
    for k in (10, 100):
        for q in (qs[1], rows[10], extra[5]):
            same_lists(sh.search_arrays(q, k), one.search_arrays(q, k), f"after rebuild k={k}")
    sh.insert_batch(ids_for(n + 400)[n + 300:], oracle.synth_rows(n + 400, d, n + 300, 100))
    one.insert_batch(ids_for(n + 400)[n + 300:], oracle.synth_rows(n + 400, d, n + 300, 100))
    same_lists(sh.search_arrays(qs[2], 50), one.search_arrays(qs[2], 50), "append after rebuild")


def _csr(lists):
    off = np.zeros(len(lists) + 1, np.uint64)
    off[1:] = np.cumsum([len(x) for x in lists])
    return off, np.array([t for x in lists for t in x], dtype=np.uint32)


@pytest.mark.parametrize("n,d,shards", [(9000, 768, 3), (6000, 100, 2)])
def test_sharded_linker_passes_equal_single_index(hip, oracle, n, d, shards):
    """cx_sharded_autolink_pass_rows / cx_sharded_dedup_scan_rows: scanned vectors scattered to every shard, per-shard
    neighbour lists merged on the root, the reference's walk (self, deleted, existing edges, caps) in global rows —
    edge for edge what one index over the same rows returns, hence what the oracle returns."""
    rows = oracle.synth_rows(n, d)
    ids = ids_for(n)
    one = hip.HipIndex(d); one.insert_batch(ids, rows)
    sh = hip.ShardedHipIndex(d, [0] * shards); sh.insert_batch(ids, rows)
    for ix in (one, sh):
        ix.remove(ids[11].tobytes()); ix.remove(ids[4100].tobytes())
    rng = np.random.default_rng(4)
    thr = float(np.float32(0.75))
    deleted = (rng.random(n) < 0.05).astype(np.uint8)
    # whole store, then an arbitrary-order subset with existing edges and a per-cycle cap
    a = sh.autolink_pass_rows(None, 100, thr, 50, deleted)
    b = one.autolink_pass_rows(None, 100, thr, 50, deleted)
    assert len(b[0]) > 1000
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    scan = rng.permutation(n)[:2500].astype(np.uint32)              # crosses a 2048-node block of the sharded pass
    first = one.autolink_pass_rows(scan, 100, float(np.float32(0.85)), 10)
    have = {}
    for f, t in zip(first[0], first[1]):
        have.setdefault(int(f), []).append(int(t))
    lists = [list(rng.permutation(have.get(int(s), []) + [int(x) for x in rng.integers(0, n, 2)])) if rng.random() < 0.3 else [] for s in scan]
    ex = _csr(lists)
    for cyc in (None, 2000):
        a = sh.autolink_pass_rows(scan, 100, thr, 10, deleted, existing=ex, max_edges_per_cycle=cyc)
        b = one.autolink_pass_rows(scan, 100, thr, 10, deleted, existing=ex, max_edges_per_cycle=cyc)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), f"cycle cap {cyc}"
    o = oracle.OracleIndex(d); o.insert_batch(ids, rows)
    o.remove(ids[11].tobytes()); o.remove(ids[4100].tobytes())
    keep = np.array([s for s in scan[:300] if s not in (11, 4100)], dtype=np.uint32)
    e = o.autolink_pass(keep, 100, thr, 10, deleted, n_threads=8)
    g = sh.autolink_pass_rows(keep, 100, thr, 10, deleted)
    same = sum(1 for x, y in zip(zip(g[0].tolist(), g[1].tolist()), zip(e["from_row"].tolist(), e["to_row"].tolist())) if x == y)
    assert len(g[0]) == len(e) and same >= 0.97 * len(e)
    # dedup: pairs reported once, by the node scanned first, in global row order
    a = sh.dedup_scan_rows(float(np.float32(0.92)), deleted)
    b = one.dedup_scan_rows(float(np.float32(0.92)), deleted)
    assert len(b[0]) > 0 and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_sharded_dedup_has_no_neighbour_cap(hip, oracle):
    """cx_sharded_dedup_scan_rows with a 400-row cluster of near-copies (more than the 256 a merged list holds): the same
    pairs, in the same order, as one index over the same rows (dedup.rs:85-87 has no k), hence as the oracle."""
    n, d, members = 3000, 384, 400
    rows = oracle.synth_rows(n, d).copy()
    rng = np.random.default_rng(21)
    where = np.sort(rng.choice(n, size=members, replace=False))
    rows[where] = rows[where[0]][None, :] + rng.normal(0.0, 0.02 / np.sqrt(d), size=(members, d)).astype(np.float32)
    ids = ids_for(n)
    one = hip.HipIndex(d); one.insert_batch(ids, rows)
    sh = hip.ShardedHipIndex(d, [0, 0, 0]); sh.insert_batch(ids, rows)
    thr = float(np.float32(0.92))
    deleted = (rng.random(n) < 0.05).astype(np.uint8)
    for dl in (None, deleted):
        a = sh.dedup_scan_rows(thr, dl)
        b = one.dedup_scan_rows(thr, dl)
        assert len(b[0]) > members * (members - 1) // 2 * 0.8
        assert np.array_equal(a[0], b[0])
        # dense rows come from the shards' scan kernels, the others from the rescore kernel: same pairs, scores within the tolerance
        pa = sorted(zip(a[0].tolist(), a[1].tolist())); pb = sorted(zip(b[0].tolist(), b[1].tolist()))
        assert pa == pb
        sa = dict(zip(zip(a[0].tolist(), a[1].tolist()), a[2].tolist())); sb = dict(zip(zip(b[0].tolist(), b[1].tolist()), b[2].tolist()))
        assert max(abs(sa[k] - sb[k]) for k in sa) <= 5e-5
    o = oracle.OracleIndex(d); o.insert_batch(ids, rows)
    e = o.dedup_scan(thr, None)
    g = sh.dedup_scan_rows(thr, None)
    assert sorted(zip(g[0].tolist(), g[1].tolist())) == sorted(zip(e["from_row"].tolist(), e["to_row"].tolist()))


def test_sharded_save_and_load_are_the_single_index_file(hip, oracle, tmp_path):
    """vector/index.rs:437-473 for the sharded handle: ONE file in the reference's layout.  A 3-shard index with removed
    rows, metadata (also for an id that has no vector yet) and a second batch of inserts saves the bytes a single index over
    the same calls saves; each kind of index loads the other's file and then answers like it."""
    n, d = 5000, 384
    rows = oracle.synth_rows(n, d); ids = ids_for(n)
    one = hip.HipIndex(d); sh = hip.ShardedHipIndex(d, [0, 0, 0])
    for ix in (one, sh):
        ix.insert_batch(ids[:4000], rows[:4000])
        for r in (3, 77, 2048): ix.remove(ids[r].tobytes())
        for r in range(0, 900, 7): ix.set_metadata(ids[r].tobytes(), "fact" if r % 2 else "event", "kai")
        ix.set_metadata(ids[4500].tobytes(), "decision", "nova")          # metadata before the vector (vector/tests.rs:65-66)
        ix.insert_batch(ids[4000:], rows[4000:])
        ix.set_metadata(ids[4999].tobytes(), "goal", "kai")
    f1, f2 = tmp_path / "one.idx", tmp_path / "sharded.idx"
    one.save(f1); sh.save(f2)
    assert open(f1, "rb").read() == open(f2, "rb").read()
    back_sh = hip.ShardedHipIndex.load(f1, [0, 0])                         # the single index's file, on two shards
    back_one = hip.HipIndex.load(f2)                                       # the sharded file, on one GPU
    assert len(back_sh) == len(back_one) == len(one) == n - 3
    flt = hip.VectorFilter(kinds=["fact", "decision"])
    for q in oracle.synth_queries(n, d, 6):
        for f in (None, flt):
            a, b, c = one.search_arrays(q, 20, f), back_sh.search_arrays(q, 20, f), back_one.search_arrays(q, 20, f)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[0], c[0])
            assert np.allclose(a[1], b[1], atol=1e-6) and np.allclose(a[1], c[1], atol=1e-6)
    with pytest.raises(hip.CortexError):
        hip.ShardedHipIndex.load(tmp_path / "missing.idx", [0])


def test_failed_shard_append_leaves_the_handle_consistent(hip):
    """Round-2 ADVICE: a shard append that fails (a hipMalloc in grow_rows is enough; here injected with
    CX_SHARD_FAIL_UPSERT in the test-hooks build of the library) must not leave ids live in the handle but absent from the shard.  The failed call reports the
    error, the rows placed before it stay, a retry of the SAME batch places the rest, and the index then answers like a
    single index over the same rows.  The switch is read once per process: a child process."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import cortex_amd
from oracle import oracle as O
O.build()
n, d = 3000, 384
rows = O.synth_rows(n, d)
rng = np.random.default_rng(1000); ids = rng.integers(0, 256, size=(n, 16), dtype=np.uint8)
ids[:, 8:] = np.arange(n, dtype=np.uint64).astype(">u8").view(np.uint8).reshape(n, 8)
sh = cortex_amd.ShardedHipIndex(d, [0, 0, 0])
try:
    sh.insert_batch(ids, rows)                  # 6 placement blocks of 512: the third append fails
    print("NOERROR")
except cortex_amd.CortexError as e:
    assert "injected failure" in str(e), str(e)
assert len(sh) == 1024 and sh.row_count() == 1024, (len(sh), sh.row_count())
sh.insert_batch(ids, rows)                      # the retry: known ids rewrite in place, the rest is appended
assert len(sh) == n and sh.row_count() == n
one = cortex_amd.HipIndex(d); one.insert_batch(ids, rows)
for q in (rows[5], rows[1500], rows[2999]):
    a, b = sh.search_arrays(q, 20), one.search_arrays(q, 20)
    assert np.array_equal(a[0], b[0]) and np.allclose(a[1], b[1], atol=1e-6)
fr = sh.autolink_pass_rows(None, 100, float(np.float32(0.85)), 50)
fo = one.autolink_pass_rows(None, 100, float(np.float32(0.85)), 50)
assert len(fo[0]) > 100 and np.array_equal(fr[0], fo[0]) and np.array_equal(fr[1], fo[1])
print("OK")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hooks = os.path.join(root, "cortex_amd", "lib", "libcortex_hip_testhooks.so")   # the fault injection is not compiled into the product
    assert os.path.exists(hooks), "build() makes libcortex_hip_testhooks.so beside the product library"
    env = dict(os.environ, CX_SHARD_FAIL_UPSERT="2", CX_SHARD_PLACEMENT_BLOCK="512", CORTEX_HIP_LIB=hooks)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == "OK", (r.stdout[-500:], r.stderr[-2000:])


def test_linker_passes_do_not_leak_per_stream_scratch(hip, oracle):
    """Round-2 ADVICE: the sharded linker passes ran on fresh streams every call, and every shard keeps one scratch context
    per stream it has been called on: a pass per linker cycle leaked HBM without bound.  The streams now live in the
    handle: free device memory stays flat over repeated passes."""
    import torch
    n, d = 6000, 384
    rows = oracle.synth_rows(n, d)
    sh = hip.ShardedHipIndex(d, [0, 0, 0]); sh.insert_batch(ids_for(n), rows)
    scan = np.arange(0, 3000, dtype=np.uint32)
    for _ in range(3):
        sh.topk_lists_rows(20, scan)
        sh.autolink_pass_rows(scan, 100, float(np.float32(0.85)), 50)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(25):
        sh.topk_lists_rows(20, scan)
        sh.autolink_pass_rows(scan, 100, float(np.float32(0.85)), 50)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < (8 << 20), f"{(free0 - free1) >> 20} MiB of device memory gone after 25 more passes"


def test_sharded_concurrent_readers_and_empty_shards(hip, oracle):
    d = 384
    sh = hip.ShardedHipIndex(d, [0, 0, 0, 0])
    assert sh.search_arrays(np.ones(d, np.float32), 5)[1].size == 0     # empty index
    rows = oracle.synth_rows(5000, d)
    ids = ids_for(5000)
    sh.insert_batch(ids[:100], rows[:100])                                # three of the four shards still empty
    one = hip.HipIndex(d); one.insert_batch(ids[:100], rows[:100])
    same_lists(sh.search_arrays(rows[3], 10), one.search_arrays(rows[3], 10), "one populated shard")
    same_lists(sh.search_arrays(rows[3], 150), one.search_arrays(rows[3], 150), "k > rows")
    sh.insert_batch(ids[100:], rows[100:]); one.insert_batch(ids[100:], rows[100:])
    qs = oracle.synth_queries(5000, d, 12)
    want = [one.search_arrays(q, 10) for q in qs]
    wantb = one.search_batch_arrays(qs, 10)
    errs = []

    def work(t):
        try:
            for rep in range(5):
                if t % 2:
                    for i in range(len(qs)):
                        same_lists(sh.search_arrays(qs[i], 10), want[i], f"thread {t}")
                else:
                    a = sh.search_batch_arrays(qs, 10)
                    assert np.array_equal(a[0], wantb[0]) and np.array_equal(a[3], wantb[3])
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs[0]
    with pytest.raises(hip.ValidationError):
        sh.profile_enable(True)                                           # an entry point with no sharded form fails loudly


def test_sharded_lists_bulk_load_and_decayed_search(hip, oracle, monkeypatch):
    """The rest of the single-index surface on the multi-GPU index: the linker's ordered top-k lists, the start-up load from
    stored `Node` records (serve.rs:105-123) and the HTTP handler's decayed search (routes.rs:889-947) — each equal to
    the single index over the same input."""
    import test_hip_bulk_load as BL
    from cortex_amd import scoring as S
    monkeypatch.setenv("CX_SHARD_PLACEMENT_BLOCK", "512")        # several placement blocks per shard at this size
    n, d = 4000, 384
    recs, nodes = BL._records(oracle, n, d, seed=5)
    one = hip.HipIndex(d)
    sh = hip.ShardedHipIndex(d, [0, 0, 0])
    st1 = one.bulk_load_nodes(recs, set_metadata=True, set_stats=True)
    st2 = sh.bulk_load_nodes(recs, set_metadata=True, set_stats=True)
    assert st1 == st2 and st2["indexed"] > 2500
    assert len(sh) == len(one) and all(sh.shard_len(i) > 0 for i in range(3))
    qs = oracle.synth_queries(n, d, 8)
    for q in qs[:4]:
        same_lists(sh.search_arrays(q, 20), one.search_arrays(q, 20), "after bulk load")
        same_lists(sh.search_arrays(q, 20, hip.VectorFilter(kinds=["decision"], source_agent="agent-1")),
                   one.search_arrays(q, 20, hip.VectorFilter(kinds=["decision"], source_agent="agent-1")), "metadata from the records")
    # decayed search: the node stats came with the records
    cfg = S.ScoreDecayConfig()
    now = (1_704_067_200 + 90 * 86400, 0)
    for q in qs:
        for limit, rb in ((10, None), (5, 0.5), (3, 1.0)):
            assert sh.search_decayed(q, limit, cfg, recency_bias=rb, now=now) == one.search_decayed(q, limit, cfg, recency_bias=rb, now=now)
    # ordered top-k lists of scanned rows (global rows), removed rows included in the scan
    for ix in (one, sh):
        ix.remove(ix.row_id(17).bytes)
    scan = np.array([0, 5, 17, 999, 2000, one.row_count() - 1] + list(range(100, 400, 7)), dtype=np.uint32)
    a = sh.topk_lists_rows(50, scan)
    b = one.topk_lists_rows(50, scan)
    assert np.array_equal(a[2], b[2]) and a[2][2] == 0
    for p in range(len(scan)):
        c = int(b[2][p])
        assert_topk_parity(a[0][p, :c].astype(np.int64), a[1][p, :c], b[0][p, :c].astype(np.int64), b[1][p, :c], tol=1e-5, what=f"lists row {scan[p]}")
    a = sh.topk_lists_rows(100, None)
    b = one.topk_lists_rows(100, None)        # >= 256 rows: the single index takes its filter path, the shards their batched search
    assert np.array_equal(a[2], b[2])
    for p in range(0, one.row_count(), 97):
        c = int(b[2][p])
        assert_topk_parity(a[0][p, :c].astype(np.int64), a[1][p, :c], b[0][p, :c].astype(np.int64), b[1][p, :c], tol=1e-5, what=f"all-rows lists row {p}")
