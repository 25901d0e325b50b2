#!/usr/bin/env python3
"""Randomised differential run of the search paths against the CPU oracle (test infrastructure): random corpus
sizes, dims, k, query counts, tombstones and filters through search / search_batch for a given number of seconds.
Exits non-zero on the first disagreement, printing the case so that it can be added to the parity tests."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cortex_amd as hip
from oracle import oracle as O
from conftest import assert_topk_parity, ids_for

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120.0)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--big", action="store_true", help="corpora of 50k-300k rows at 384/768-d: many compactions per list, several query groups")
ap.add_argument("--dtype", default="f32", help="bf16: the index is a bf16 row store (cx_create_ex) and the oracle is fed the rounded rows")
ap.add_argument("--selective", type=float, default=0.0, help="share of the cases with a SELECTIVE filter (every row with metadata, one row in 7 / 50 / 500 passes); "
                "0 keeps the random stream of the fixed-seed runs in tests/test_hip_fuzz.py as it always was")
ap.add_argument("--only-case", type=int, default=None, help="replay: draw cases 0..N-1 without running them (same random stream), run case N alone and print both sides")
ap.add_argument("--irregular", type=float, default=0.25, help="share of the cases that hold vectors scaled by 1e+-20 / 1e-25, Inf or NaN elements (rows and queries)")
a = ap.parse_args()


def spoil(v, rng, count):
    """`count` vectors of v made irregular in place: |x|^2 overflows / underflows in f32, or an element is not finite"""
    v = v.copy()
    with np.errstate(over="ignore", under="ignore", invalid="ignore"):
        for r in rng.integers(0, len(v), count):
            kind = int(rng.integers(0, 6))
            if kind == 0: v[r] = v[r] * np.float32(1e20)
            elif kind == 1: v[r] = v[r] * np.float32(1e-25)
            elif kind == 2: v[r, int(rng.integers(0, v.shape[1]))] = np.inf
            elif kind == 3: v[r, int(rng.integers(0, v.shape[1]))] = np.nan
            elif kind == 4: v[r] = v[r] * np.float32(-1e-16)
            else: v[r] = 0.0
    return v


def stored(x):   # what the index keeps of a row
    if a.dtype != "bf16":
        return x
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32)).view(np.float32).reshape(x.shape)
O.build()
rng = np.random.default_rng(a.seed)
t_end = time.time() + a.seconds
cases = 0
while time.time() < t_end:
    live = a.only_case is None or cases == a.only_case
    if a.only_case is not None and cases > a.only_case: break
    d = int(rng.choice([384, 768, 384, 768, 128, 100, 1024, 64, 512, 640]))
    n = int(rng.choice([rng.integers(1, 200), rng.integers(200, 5000), rng.integers(5000, 40000)]))
    k = int(rng.choice([1, 5, 10, 16, 32, 33, 64, 100, 104, 105, 300]))
    nq = int(rng.choice([1, 2, 3, 17, 32, 33, 64, 65, 130]))
    rows = O.synth_rows(n, d, seed_rows=int(rng.integers(1, 1 << 30)))
    if rng.random() < 0.2 and n > 4:
        rows = rows.copy(); rows[rng.integers(0, n, 3)] = 0.0                  # zero-norm rows (NaN scores)
    if rng.random() < 0.2 and n > 10:
        rows = rows.copy(); rows[n // 2:] = rows[: n - n // 2]                # every row twice: ties everywhere
    irregular = rng.random() < a.irregular
    if irregular:
        rows = spoil(rows, rng, int(rng.integers(1, 5)))
    ids = ids_for(n)
    h = o = None
    if live:
        h = hip.HipIndex(d, dtype=a.dtype); h.insert_batch(ids, rows)
        o = O.OracleIndex(d); o.insert_batch(ids, stored(rows))
    removed = [int(r) for r in rng.integers(0, n, int(rng.integers(0, 6)))]
    for r in removed:
        if live: h.remove(ids[r].tobytes()); o.remove(ids[r].tobytes())
    def selective_filter(h, o, ids, n):
        """every row with metadata, one row in S of a kind the filter asks for: bounds from a handful of passing rows (or none)"""
        S, s0 = int(rng.choice([7, 50, 500])), int(rng.integers(0, 7))
        if not live: return None, None
        kinds_all = ["sparse" if r % S == s0 else "common" for r in range(n)]
        h.set_metadata_batch(ids, kinds_all, ["kai"] * n)
        for r in range(n): o.set_metadata(ids[r].tobytes(), kinds_all[r], "kai")
        return hip.VectorFilter(kinds=["sparse"]), O.Filter(kinds=["sparse"])
    hf = of = None
    fmode = rng.random()
    if fmode < a.selective:
        hf, of = selective_filter(h, o, ids, n)
    elif fmode < (0.4 if a.selective > 0.0 else 0.3):
        for r in range(0, n, 2):
            if live: h.set_metadata(ids[r].tobytes(), "fact" if r % 4 else "event", "kai"); o.set_metadata(ids[r].tobytes(), "fact" if r % 4 else "event", "kai")
        ex = [ids[int(i)].tobytes() for i in rng.integers(0, n, 4)]
        if live: hf, of = hip.VectorFilter(kinds=["fact"], exclude=ex), O.Filter(kinds=["fact"], exclude=ex)
    if a.big:
        d = int(rng.choice([384, 768]))
        n = int(rng.integers(50_000, 300_000))
        k = int(rng.choice([1, 10, 32, 33, 100, 104]))
        nq = int(rng.choice([33, 64, 65, 130]))
        rows = O.synth_rows(n, d, seed_rows=int(rng.integers(1, 1 << 30)))
        if irregular:
            rows = spoil(rows, rng, int(rng.integers(1, 5)))
        ids = ids_for(n)
        if live:
            h = hip.HipIndex(d, dtype=a.dtype); h.insert_batch(ids, rows)
            o = O.OracleIndex(d); o.insert_batch(ids, stored(rows))
        for r in rng.integers(0, n, 5):
            if live: h.remove(ids[r].tobytes()); o.remove(ids[r].tobytes())
        hf = of = None
        if a.selective > 0.0 and fmode < 0.4:
            hf, of = selective_filter(h, o, ids, n)
    qs = O.synth_queries(max(n, 64), d, nq, seed_centres=int(rng.integers(1, 1 << 30)))
    if irregular and rng.random() < 0.7:
        qs = spoil(qs, rng, int(rng.integers(1, 3)))
    if not live:
        cases += 1
        continue
    lut = {ids[i].tobytes(): i for i in range(n)}
    if a.only_case is not None:   # replay: both sides of the first queries, and what the rows are
        with np.errstate(over="ignore", invalid="ignore"):
            print("case", cases, "n", n, "d", d, "k", k, "nq", nq, "removed", removed, "filter", hf is not None)
            print("row sums of squares:", [float(np.sum(np.float32(r) * np.float32(r), dtype=np.float32)) for r in rows[:min(n, 8)]], "non-finite rows:", [i for i in range(n) if not np.all(np.isfinite(rows[i]))])
            print("query sums of squares:", [float(np.sum(q * q, dtype=np.float32)) for q in qs[:min(nq, 4)]], "non-finite queries:", [i for i in range(nq) if not np.all(np.isfinite(qs[i]))])
        bi, bs, bd, bc = h.search_batch_arrays(qs, k, hf)
        for i in range(min(nq, 6)):
            e = o.search(qs[i], k, of); m = int(bc[i])
            gi, gs, gd = h.search_arrays(qs[i], k, hf)
            print(f"q{i}: batch rows", [lut[x.tobytes()] for x in bi[i, :m]], "scores", list(bs[i, :m]), "| single rows", [lut[x.tobytes()] for x in gi], "scores", list(gs),
                  "| oracle rows", list(e["row"]), "scores", list(e["score"]))
    what = f"case n={n} d={d} k={k} nq={nq} filter={hf is not None} dtype={a.dtype} irregular={irregular}"
    try:
        bi, bs, bd, bc = h.search_batch_arrays(qs, k, hf)
        exp_all = o.search_batch(qs, k, of, n_threads=16) if a.big else None
        for i in range(nq):
            e = exp_all[i] if exp_all is not None else o.search(qs[i], k, of)
            m = int(bc[i])
            assert m == len(e["row"]), f"{what}: q{i} count {m} != {len(e['row'])}"
            assert_topk_parity(np.array([lut[x.tobytes()] for x in bi[i, :m]]), bs[i, :m], e["row"], e["score"], what=f"{what} batch q{i}")
        for i in range(min(nq, 3)):
            gi, gs, gd = h.search_arrays(qs[i], k, hf)
            e = o.search(qs[i], k, of)
            assert_topk_parity(np.array([lut[x.tobytes()] for x in gi]), gs, e["row"], e["score"], what=f"{what} single q{i}")
    except AssertionError as err:
        print("MISMATCH", what, "seed", a.seed, "after", cases, "cases:", err)
        sys.exit(1)
    cases += 1
print(f"{cases} random cases agree with the oracle (seed {a.seed}, {a.seconds:.0f} s)")
