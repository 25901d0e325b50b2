#!/usr/bin/env python3
"""Randomised STATEFUL differential run (test infrastructure): a random sequence of insert / upsert / remove /
set_metadata / rebuild on a GPU index and on the CPU oracle side by side, with searches (single, batched, threshold),
top-k lists and dedup scans checked in between.  Ids are compared, not rows: after a rebuild the two sides number
their rows alike only by construction, which is part of what is checked.  Exits non-zero on the first difference."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cortex_amd as hip
from oracle import oracle as O
from conftest import assert_topk_parity, ids_for, SCORE_TOL

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120.0)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--dtype", default="f32", help="bf16: a bf16 row store (cx_create_ex); the oracle is fed the rounded rows")
a = ap.parse_args()


def stored(x):   # what the index keeps of a row
    if a.dtype != "bf16":
        return x
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32)).view(np.float32).reshape(np.shape(x))
O.build()
rng = np.random.default_rng(a.seed)
t_end = time.time() + a.seconds
runs = ops = 0
while time.time() < t_end:
    d = int(rng.choice([384, 768, 128]))
    pool_n = int(rng.integers(50, 6000))
    pool = O.synth_rows(pool_n, d, seed_rows=int(rng.integers(1, 1 << 30)))
    pids = ids_for(pool_n, salt=int(rng.integers(0, 1000)))
    h, o = hip.HipIndex(d, dtype=a.dtype), O.OracleIndex(d)
    live = {}                       # id bytes -> current vector
    log = []

    def check(tag):
        if not live:
            assert h.len() == 0 == len(o)
            return
        assert h.len() == len(o) == len(live), f"{tag}: len {h.len()} / {len(o)} / {len(live)}"
        k = int(rng.choice([1, 10, 40, 100]))
        qs = O.synth_queries(max(pool_n, 64), d, int(rng.choice([1, 5, 40, 70])), seed_centres=int(rng.integers(1, 1 << 30)))
        hf = of = None
        if rng.random() < 0.3:
            ex = [pids[int(i)].tobytes() for i in rng.integers(0, pool_n, 3)]
            hf, of = hip.VectorFilter(kinds=["fact"], exclude=ex), O.Filter(kinds=["fact"], exclude=ex)
        bi, bs, bd, bc = h.search_batch_arrays(qs, k, hf)
        for i in range(len(qs)):
            e = o.search(qs[i], k, of)
            m = int(bc[i])
            assert m == len(e), f"{tag}: batch q{i} count {m} vs {len(e)}"
            # compare by id: positions where adjacent oracle scores are SCORE_TOL apart may swap
            got_ids = [bi[i, j].tobytes() for j in range(m)]
            exp_ids = [bytes(x) for x in e["node_id"]]
            if got_ids != exp_ids:
                es = e["score"].astype(np.float64)
                for j, (g, x) in enumerate(zip(got_ids, exp_ids)):
                    if g != x:
                        near = [t for t in range(m) if abs(es[t] - es[j]) <= SCORE_TOL or (np.isnan(es[t]) and np.isnan(es[j]))]
                        assert g in [exp_ids[t] for t in near] or abs(bs[i, j] - es[min(j, m - 1)]) <= SCORE_TOL, f"{tag}: batch q{i} pos {j} differs beyond near-ties"
            ok = ~np.isnan(e["score"])
            assert np.all(np.abs(bs[i, :m][ok] - e["score"][ok]) <= SCORE_TOL), f"{tag}: batch q{i} scores"
        gi, gs, gd = h.search_threshold_arrays(qs[0], 0.8, hf)
        e = o.search_threshold(qs[0], np.float32(0.8), of)
        near = set(bytes(x) for x in o.search_threshold(qs[0], np.float32(0.8 - SCORE_TOL), of)["node_id"]) - \
            set(bytes(x) for x in o.search_threshold(qs[0], np.float32(0.8 + SCORE_TOL), of)["node_id"])
        diff = set(x.tobytes() for x in gi) ^ set(bytes(x) for x in e["node_id"])
        assert diff <= near, f"{tag}: threshold search differs beyond the threshold band ({len(diff)} ids)"

    try:
        for step in range(int(rng.integers(5, 40))):
            op = rng.random()
            if op < 0.45 or not live:
                m = int(rng.integers(1, 400))
                idx = rng.integers(0, pool_n, m)
                idx = np.unique(idx)
                if rng.random() < 0.5:
                    h.insert_batch(pids[idx], pool[idx]); o.insert_batch(pids[idx], stored(pool[idx]))
                else:
                    for i in idx[:50]:
                        h.insert(pids[i].tobytes(), pool[i]); o.insert(pids[i].tobytes(), stored(pool[i]))
                    idx = idx[:50]
                for i in idx: live[pids[i].tobytes()] = i
                log.append(("insert", len(idx)))
            elif op < 0.6:
                keys = list(live)
                for kk in [keys[int(t)] for t in rng.integers(0, len(keys), min(len(keys), int(rng.integers(1, 30))))]:
                    j = int(rng.integers(0, pool_n))                     # upsert: a known id gets another vector, keeps its row
                    h.insert(kk, pool[j]); o.insert(kk, stored(pool[j])); live[kk] = j
                log.append(("upsert",))
            elif op < 0.8:
                keys = list(live)
                for kk in set(keys[int(t)] for t in rng.integers(0, len(keys), min(len(keys), int(rng.integers(1, 60))))):
                    h.remove(kk); o.remove(kk); live.pop(kk)
                log.append(("remove",))
            elif op < 0.9:
                keys = list(live)
                for kk in [keys[int(t)] for t in rng.integers(0, len(keys), min(len(keys), 40))]:
                    kind = "fact" if rng.random() < 0.5 else "event"
                    h.set_metadata(kk, kind, "kai"); o.set_metadata(kk, kind, "kai")
                log.append(("meta",))
            else:
                h.rebuild(); o.rebuild()
                log.append(("rebuild",))
            ops += 1
            if rng.random() < 0.5:
                check(f"seed {a.seed} run {runs} d={d} after {log[-6:]}")
        check(f"seed {a.seed} run {runs} d={d} final")
    except AssertionError as err:
        print("MISMATCH", err)
        sys.exit(1)
    runs += 1
print(f"{runs} random histories ({ops} mutations) agree with the oracle (seed {a.seed}, {a.seconds:.0f} s)")
