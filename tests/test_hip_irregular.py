"""Non-finite and extreme-magnitude vectors through every path, against the oracle (= the reference's arithmetic).

`insert` validates the length only (vector/index.rs:298-314), so a store may hold anything, and `EmbeddingPoint::distance`
(:169-179) is plain f32 arithmetic whatever it is fed:
  * a row scaled by 1e20: its squares sum to +inf -> dot / (|q| * inf) = 0 -> score 0 (NaN if the dot overflowed too);
  * a row scaled by 1e-25: every square underflows to 0 -> dot / 0 = +-inf -> distance -+inf -> score 1.0 or 0.0;
  * a row with an Inf or a NaN element, the zero row: NaN (sorted last: `partial_cmp -> Equal`, :287-291; order = conftest);
  * the same for queries.
The screening paths (batchs.hip, the filter GEMM) work on L2-normalised bf16 copies, where none of this exists: such
vectors are IRREGULAR (kernels.hpp: bs_regular) — zero shadow rows that no screening pass lists, carried along as
candidates of every query / scanned row and scored with the reference's arithmetic; irregular queries are redone
exactly.  Each path below must return what `cxo_*` returns: ids exact, scores equal (NaN = NaN), distances equal
(+-inf = +-inf)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import numpy as np, sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import cortex_amd
from oracle import oracle
from conftest import SCORE_TOL, assert_topk_parity, ids_for
oracle.build()
d, dtype, n = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
def rnd(x):
    if dtype != "bf16": return x
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    r = np.where(nan, (u & 0xFFFF0000) | 0x00400000, r)
    return r.astype(np.uint32).view(np.float32).reshape(x.shape)
rows = oracle.synth_rows(n, d)
with np.errstate(over="ignore", under="ignore"):
    BIG, TINY, PINF, NAN, ZERO, NINF, BIG2, SMALL_OK, SMALL_IRR, LARGE_OK = 10, 11, 40, 41, 42, 43, n - 2, 100, 101, 102
    rows[BIG] = rows[BIG] * np.float32(1e20)
    rows[BIG2] = rows[BIG2] * np.float32(-1e20)
    rows[TINY] = rows[TINY] * np.float32(1e-25)
    rows[PINF, 5] = np.inf
    rows[NINF, d - 1] = -np.inf
    rows[NAN, 7] = np.nan
    rows[ZERO] = 0.0
    rows[SMALL_OK] = rows[SMALL_OK] * np.float32(1e-12)     # |x|^2 = 1e-24: regular
    rows[SMALL_IRR] = rows[SMALL_IRR] * np.float32(1e-16)   # |x|^2 = 1e-32: exact in the reference, outside the screening range
    rows[LARGE_OK] = rows[LARGE_OK] * np.float32(1e12)
    qs = oracle.synth_queries(n, d, 12)
    qs[6] = rows[TINY]                  # an irregular row as its own query (the linker's search of that node)
    qs[7, 3] = np.nan
    qs[8] = 0.0
    qs[9] = qs[9] * np.float32(1e20)
    qs[10] = qs[10] * np.float32(1e-25)
    qs[11, 0] = np.inf
ids = ids_for(n); lut = {ids[i].tobytes(): i for i in range(n)}
h = cortex_amd.HipIndex(d, dtype=dtype); h.insert_batch(ids, rows)
o = oracle.OracleIndex(d); o.insert_batch(ids, rnd(rows))

def same_dist(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    fin = np.isfinite(a) & np.isfinite(b)
    return bool(np.all(np.where(fin, np.abs(a - b) <= SCORE_TOL, (np.isnan(a) & np.isnan(b)) | (a == b))))
def check_list(got_ids, gs, gd, e, what):
    m = len(e["row"])
    assert len(gs) == m, (what, len(gs), m)
    got = np.array([lut[g.tobytes()] for g in got_ids], dtype=np.int64)
    assert_topk_parity(got, gs, e["row"], e["score"], what=what)
    if list(map(int, got)) == list(map(int, e["row"])):
        assert same_dist(gd, e["distance"]), (what, gd[:6], e["distance"][:6])

# 1. single-query scans, every k class, and the threshold search
for k in (1, 10, 100, 300):
    for i, q in enumerate(qs):
        gi, gs, gd = h.search_arrays(q, k)
        check_list(gi, gs, gd, o.search(q, k), "scan k=%%d q%%d" %% (k, i))
for i, q in enumerate(qs):
    for thr in (0.5, 1.0):
        gi, gs, gd = h.search_threshold_arrays(q, thr)
        check_list(gi, gs, gd, o.search_threshold(q, thr), "threshold %%g q%%d" %% (thr, i))
# 2. the batched search (whichever kernel the environment of this process routes to)
for k in (10, 100):
    bi, bs, bd, bc = h.search_batch_arrays(qs, k)
    for i in range(len(qs)):
        m = int(bc[i])
        check_list(bi[i, :m], bs[i, :m], bd[i, :m], o.search(qs[i], k), "batch k=%%d q%%d" %% (k, i))
# more than 64 queries in one call (row widths up to 512 on stores of >= 28,672 rows: 128 queries per pass, two banks of the
# screening kernel — irregular queries in both banks)
qs_many = np.ascontiguousarray(np.concatenate([qs, qs[::-1], oracle.synth_queries(n, d, 70), qs[5:]], axis=0))
bi, bs, bd, bc = h.search_batch_arrays(qs_many, 10)
for i in range(0, len(qs_many), 3):
    m = int(bc[i])
    check_list(bi[i, :m], bs[i, :m], bd[i, :m], o.search(qs_many[i], 10), "many queries q%%d" %% i)
# ... again after an irregular row was replaced by a regular one and a regular one by an irregular one (in-place upserts:
# the screening copy and the irregular list follow), and with a filter + tombstones
with np.errstate(over="ignore", under="ignore"):
    v1 = oracle.synth_rows(n, d)[PINF]; v2 = (rows[200] * np.float32(1e-25)).astype(np.float32)
h.insert(ids[PINF].tobytes(), v1); o.insert(ids[PINF].tobytes(), rnd(v1))
h.insert(ids[200].tobytes(), v2); o.insert(ids[200].tobytes(), rnd(v2))
h.insert(ids[TINY].tobytes(), rows[TINY]); o.insert(ids[TINY].tobytes(), rnd(rows[TINY]))   # (the same irregular row again: listed once)
rows[PINF] = v1; rows[200] = v2
for r in range(0, n, 3):
    kind = "fact" if r %% 2 else "event"
    h.set_metadata(ids[r].tobytes(), kind, "kai"); o.set_metadata(ids[r].tobytes(), kind, "kai")
for r in (BIG2, 12, 500):
    h.remove(ids[r].tobytes()); o.remove(ids[r].tobytes())
for hf, of, name in ((None, None, "tombstones"), (cortex_amd.VectorFilter(kinds=["fact"]), oracle.Filter(kinds=["fact"]), "filtered")):
    bi, bs, bd, bc = h.search_batch_arrays(qs, 10, hf)
    for i in range(len(qs)):
        m = int(bc[i])
        check_list(bi[i, :m], bs[i, :m], bd[i, :m], o.search(qs[i], 10, of), "batch after upserts, %%s q%%d" %% (name, i))
        gi, gs, gd = h.search_arrays(qs[i], 10, hf)
        check_list(gi, gs, gd, o.search(qs[i], 10, of), "scan after upserts, %%s q%%d" %% (name, i))

# 3. the linker's pass and the dedup scan on a fresh pair of indexes (no tombstones): every row scanned (the filter GEMM), a
# small scan set that holds the irregular rows (the streaming / threshold-mode filter), dedup
h2 = cortex_amd.HipIndex(d, dtype=dtype); h2.insert_batch(ids, rows)
o2 = oracle.OracleIndex(d); o2.insert_batch(ids, rnd(rows))
def per_node(fr, to, w):
    out = {}
    for a, b, s in zip(fr, to, w): out.setdefault(int(a), []).append((int(b), float(s)))
    return out
def same_edges(got, exp, what):
    assert set(got) == set(exp), (what, sorted(set(got) ^ set(exp))[:10])
    for node in exp:
        g, e = got[node], exp[node]
        if [x[0] for x in g] != [x[0] for x in e]:
            # near-threshold / near-tie pairs may swap; everything else is exact
            sc = {j: s for j, s in e}
            for j, s in g:
                assert j in sc or abs(s - thr32) <= SCORE_TOL or (e and abs(s - e[-1][1]) <= SCORE_TOL), (what, node, j, s)
        else:
            assert all(abs(a[1] - b[1]) <= SCORE_TOL for a, b in zip(g, e)), (what, node)
thr32 = float(np.float32(0.85))
big_store = n > 10000    # (the oracle's all-rows pass is n brute-force searches: a 300-row scan set — still the filter GEMM's size class — instead)
first = np.array([TINY, BIG, 41, 42, 43, 200, 100, 101, 102] + list(range(1000, 1291)), dtype=np.uint32) if big_store else None
for scan, name in ((first, "300 rows" if big_store else "all rows"), (np.array([3, TINY, BIG, 41, 42, 200, 102, 101, 100, 77, 900, 5] + list(range(300, 330)), dtype=np.uint32), "small scan set")):
    scan_o = np.arange(n, dtype=np.uint32) if scan is None else scan
    fr, to, w = h2.autolink_pass_rows(scan, 100, thr32, 50)
    e = o2.autolink_pass(scan_o, 100, thr32, 50, n_threads=8)
    got, exp = per_node(fr, to, w), per_node(e["from_row"], e["to_row"], e["weight"])
    assert TINY in exp and len(exp[TINY]) == 50, "the tiny row scores 1.0 against half the store"
    same_edges(got, exp, "autolink " + name)
if not big_store:
    a, b, s = h2.dedup_scan_rows(float(np.float32(0.92)))
    e = o2.dedup_scan(float(np.float32(0.92)))
    gp = {(int(x), int(y)): float(z) for x, y, z in zip(a, b, s)}
    ep = {(int(x), int(y)): float(z) for x, y, z in zip(e["from_row"], e["to_row"], e["weight"])}
    for key in set(gp) ^ set(ep):
        v = gp.get(key, ep.get(key))
        assert abs(v - 0.92) <= SCORE_TOL, ("dedup pair", key, v)
    assert any(TINY in key for key in ep), "the tiny row is a duplicate of every row it has a positive dot with"
# ordered top-k lists of every row (a14'), irregular rows included as rows and as neighbours
lr, ls, lc = h2.topk_lists_rows(20, np.array([0, TINY, BIG, 41, 5, 101], dtype=np.uint32))
for t, r in enumerate([0, TINY, BIG, 41, 5, 101]):
    e = o2.search(rnd(rows[r:r + 1])[0], 20)
    m = int(lc[t])
    assert m == len(e["row"]), (r, m)
    assert_topk_parity(lr[t, :m].astype(np.int64), ls[t, :m], e["row"], e["score"], what="top-k list of row %%d" %% r)
print("ok")
"""


@pytest.mark.parametrize("d,dtype,n,env", [
    (384, "f32", 3000, {}),                                                      # scans, batch2 / batchg, the filter GEMM, the 128-tile / stream filter
    (768, "f32", 3000, {"CX_BATCHS_MIN_ROWS": "256"}),                           # the screening pass + threshold mode at test sizes
    (384, "f32", 3000, {"CX_BATCHS_MIN_ROWS": "256", "CX_BATCHS_CAND_CAP": "64"}),   # ... with candidate lists that run over: the exact redo
    (1024, "bf16", 2500, {"CX_BATCHS_MIN_ROWS": "256"}),                         # a bf16 store (the oracle sees the rounded rows)
    (128, "f32", 40000, {}),                                                     # default routing into the screening pass (>= 32,768 rows)
], ids=["default-384", "screening-768", "screening-overflow-384", "screening-bf16-1024", "default-routing-128"])
def test_irregular_vectors_through_every_path(d, dtype, n, env):
    code = CODE % (ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code, str(d), dtype, str(n)], capture_output=True, text=True, timeout=1500,
                       env=dict(os.environ, **env))
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-6000:]


def test_more_irregular_rows_than_a_pass_carries_switches_the_screening_off():
    """BS_IRR_CAP = 1,024 irregular rows are carried along by the screening passes; a store with more of them (bulk-loaded
    zero vectors, say) is served by the kernels that read the stored rows — same results."""
    code = r"""
import numpy as np, sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import cortex_amd
from oracle import oracle
from conftest import assert_topk_parity, ids_for
oracle.build()
n, d = 6000, 384
rows = oracle.synth_rows(n, d)
rows[::4] = 0.0                      # 1,500 zero rows
rows[1] = rows[1] * np.float32(1e-25)
qs = oracle.synth_queries(n, d, 70)
ids = ids_for(n); lut = {ids[i].tobytes(): i for i in range(n)}
h = cortex_amd.HipIndex(d); h.insert_batch(ids, rows)
o = oracle.OracleIndex(d); o.insert_batch(ids, rows)
for k in (10, 100):
    bi, bs, bd, bc = h.search_batch_arrays(qs, k)
    for i in range(len(qs)):
        e = o.search(qs[i], k); m = int(bc[i])
        assert m == len(e["row"])
        assert_topk_parity(np.array([lut[g.tobytes()] for g in bi[i, :m]]), bs[i, :m], e["row"], e["score"], what="k=%%d q%%d" %% (k, i))
thr = float(np.float32(0.85))
fr, to, w = h.autolink_pass_rows(None, 100, thr, 50)
e = o.autolink_pass(np.arange(n), 100, thr, 50, n_threads=8)
assert len(fr) == len(e) and list(map(int, fr)) == [int(x) for x in e["from_row"]]
assert [int(x) for x in to] == [int(x) for x in e["to_row"]] or abs(len(to) - len(e)) == 0
print("ok")
""" % (ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, CX_BATCHS_MIN_ROWS="256"))
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-6000:]


def _store_with_a_row_of_denormals(hip, oracle, d, n=3000):
    import numpy as np
    from conftest import ids_for
    rows = oracle.synth_rows(n, d).copy()
    with np.errstate(under="ignore"):
        rows[5] = (rows[5] * np.float32(1e-25)) * np.float32(1e-16)   # elements ~1e-43: f32 denormals, squares all 0
    rows[7] = 0.0
    ids = ids_for(n)
    h = hip.HipIndex(d); h.insert_batch(ids, rows)
    o = oracle.OracleIndex(d); o.insert_batch(ids, rows)
    qs = oracle.synth_queries(n, d, 8)
    return h, o, ids, qs, n


@pytest.mark.parametrize("d", [512, 384])
def test_a_row_of_f32_denormals_single_query_scan(hip, oracle, d):
    """A row whose ELEMENTS are f32 denormals (found by scripts/fuzz_parity.py --seed 97 --irregular 0.4, case 177): its squares
    are all 0 and its dot with a query is a denormal, so the reference computes +-denormal / 0 = +-inf -> score 1.0 or 0.0
    (vector/index.rs:172-177).  The single-query scan is plain f32 arithmetic with denormals kept: equal to the oracle."""
    import numpy as np
    from conftest import assert_topk_parity
    h, o, ids, qs, n = _store_with_a_row_of_denormals(hip, oracle, d)
    lut = {ids[i].tobytes(): i for i in range(n)}
    for i in range(len(qs)):
        gi, gs, gd = h.search_arrays(qs[i], n)
        e = o.search(qs[i], n)
        got = np.array([lut[x.tobytes()] for x in gi])
        assert_topk_parity(got, gs, e["row"], e["score"], what=f"denormal row, single q{i}")
        j = list(got).index(5)
        assert gs[j] in (0.0, 1.0) and np.isnan(gs[list(got).index(7)])


@pytest.mark.parametrize("d", [512, 384])
def test_a_row_of_f32_denormals_small_store_batch_kernels(hip, oracle, d):
    """batch.hip / batchg.hip split rows into bf16 terms, where an f32 denormal is 0 — such a row's dot came out 0 and its score NaN
    instead of the reference's 0.0 / 1.0.  A store that holds a LOSSY row (internal.hpp: found while the norms are taken) takes the
    per-query scans for its batches; a store without one keeps the batch kernels (the timing below)."""
    import time
    import numpy as np
    from conftest import assert_topk_parity
    h, o, ids, qs, n = _store_with_a_row_of_denormals(hip, oracle, d)
    lut = {ids[i].tobytes(): i for i in range(n)}
    k = 10   # (the top-k kernels; k = n would take the sort path, plain f32 like the scan.  The row scores 1.0 — first — for every
    #          query whose dot with it is a positive denormal)
    bi, bs, bd, bc = h.search_batch_arrays(qs, k)
    assert any(int(o.search(q, k)["row"][0]) == 5 for q in qs)
    for i in range(len(qs)):
        e = o.search(qs[i], k)
        m = int(bc[i])
        assert m == len(e["row"])
        assert_topk_parity(np.array([lut[x.tobytes()] for x in bi[i, :m]]), bs[i, :m], e["row"], e["score"], what=f"denormal row, batch q{i}")
    # the same store without the row: the batch kernels again (64 queries in one call well under 64 scans' time)
    from conftest import ids_for
    rows = oracle.synth_rows(n, d)
    h2 = hip.HipIndex(d); h2.insert_batch(ids_for(n), rows)
    q64 = oracle.synth_queries(n, d, 64)
    h2.search_batch_arrays(q64, k); h.search_batch_arrays(q64, k)
    t0 = time.perf_counter(); h2.search_batch_arrays(q64, k); t_fast = time.perf_counter() - t0
    t0 = time.perf_counter(); h.search_batch_arrays(q64, k); t_scans = time.perf_counter() - t0
    assert t_fast < t_scans, (t_fast, t_scans)
