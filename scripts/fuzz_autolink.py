#!/usr/bin/env python3
"""Randomised differential run of the auto-link pass (all rows / a scan subset, thresholds, per-node caps, storage
tombstones, removed rows, several dims) and of the dedup scan against the CPU oracle's restatement of the reference loop
(test infrastructure).  Exits non-zero on the first unexplained difference."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cortex_amd as hip
from oracle import oracle as O
from conftest import ids_for
from test_hip_autolink import compare_edges, per_node, oracle_scores

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
cases = 0
while time.time() < t_end:
    d = int(rng.choice([768, 384, 768, 384, 1024, 512, 128, 100]))
    n = int(rng.choice([rng.integers(2, 300), rng.integers(300, 1500), rng.integers(1500, 3500)]))
    thr = float(np.float32(rng.choice([0.85, 0.75, 0.92, 0.5, 0.3])))
    cap_e = int(rng.choice([50, 50, 5, 1, 200]))
    topk = int(rng.choice([100, 100, 10, 256]))
    rows = O.synth_rows(n, d, seed_rows=int(rng.integers(1, 1 << 30)))
    ids = ids_for(n)
    h = hip.HipIndex(d, dtype=a.dtype); h.insert_batch(ids, rows)
    rows = stored(rows)                       # from here on "the rows" are what the index holds
    o = O.OracleIndex(d); o.insert_batch(ids, rows)
    for r in rng.integers(0, n, int(rng.integers(0, 4))):
        h.remove(ids[r].tobytes()); o.remove(ids[r].tobytes())
    deleted = None
    if rng.random() < 0.4:
        deleted = (rng.random(n) < 0.1).astype(np.uint8)
    mode = rng.random()
    if mode < 0.5:
        scan = None
    elif mode < 0.8:
        scan = np.unique(rng.integers(0, n, int(rng.integers(1, 65)))).astype(np.uint32)       # stream / small kernels
    else:
        scan = np.unique(rng.integers(0, n, int(rng.integers(65, 600)))).astype(np.uint32)     # 128 / 256 tiles, asymmetric
    scan_o = np.arange(n, dtype=np.uint32) if scan is None else scan
    # existing edges of ~30 % of the scanned nodes (auto_linker.rs:226-231) and a per-cycle cap (:284-287), half of the cases
    existing, cyc = None, None
    if rng.random() < 0.5:
        lists = [[int(x) for x in rng.integers(0, n, int(rng.integers(1, 12)))] if rng.random() < 0.3 else [] for _ in scan_o]
        off = np.zeros(len(lists) + 1, np.uint64); off[1:] = np.cumsum([len(x) for x in lists])
        existing = (off, np.array([t for x in lists for t in x], dtype=np.uint32))
        cyc = int(rng.choice([2000, 37, 1 << 40]))
    what = (f"case n={n} d={d} thr={thr} cap={cap_e} topk={topk} scan={'all' if scan is None else len(scan)} deleted={deleted is not None} "
            f"existing={existing is not None} cycle_cap={cyc}")
    try:
        fr, to, w = h.autolink_pass_rows(scan, topk, thr, cap_e, deleted, existing=existing, max_edges_per_cycle=cyc)
        want = o.autolink_pass(scan_o, topk, np.float32(thr), cap_e, deleted, n_threads=8, existing=existing, max_edges_per_cycle=cyc)
        if cyc is not None and cyc < (1 << 40) and (len(fr) == cyc or len(want) == cyc):
            # a truncated cycle: both sides must have cut at the same length; the common prefix is compared node by node below,
            # except the node the cut fell in (near-ties may order its last edges differently)
            assert len(fr) == len(want), f"{what}: {len(fr)} vs {len(want)} edges after the per-cycle cap"
            last = int(want["from_row"][-1]) if len(want) else -1
            keep_g = np.array([int(x) != last for x in fr], bool); keep_w = np.array([int(x) != last for x in want["from_row"]], bool)
            fr, to, w, want = fr[keep_g], to[keep_g], w[keep_g], want[keep_w]
        compare_edges(per_node(fr, to, w), per_node(want["from_row"], want["to_row"], want["weight"]), thr, oracle_scores(o, rows), what)
        order = [int(x) for x in fr]
        scan_list = list(range(n)) if scan is None else [int(x) for x in scan]
        pos = {r: i for i, r in enumerate(scan_list)}
        assert all(pos[order[i]] <= pos[order[i + 1]] for i in range(len(order) - 1)), f"{what}: edges not in scan order"
        if rng.random() < 0.4:   # DedupScanner::scan (dedup.rs:65-127) over the same index
            dthr = float(np.float32(rng.choice([0.92, 0.85, 0.97])))
            pa, pb, psim = h.dedup_scan_rows(dthr, deleted)   # no neighbour cap (dedup.rs:85-87): dense rows take the threshold path
            if True:
                wd = o.dedup_scan(np.float32(dthr), deleted)
                got_p = {(int(x), int(y)): float(z) for x, y, z in zip(pa, pb, psim)}
                exp_p = {(int(e["from_row"]), int(e["to_row"])): float(e["weight"]) for e in wd}
                for key in set(got_p) ^ set(exp_p):
                    sc = got_p.get(key, exp_p.get(key))
                    assert abs(sc - dthr) <= 5e-5, f"{what}: dedup pair {key} score {sc} (thr {dthr})"
                for key in set(got_p) & set(exp_p):
                    assert abs(got_p[key] - exp_p[key]) <= 5e-5, f"{what}: dedup pair {key} similarity"
    except AssertionError as err:
        print("MISMATCH", what, "seed", a.seed, "after", cases, "cases:", err)
        sys.exit(1)
    cases += 1
print(f"{cases} random auto-link cases agree with the oracle (seed {a.seed}, {a.seconds:.0f} s)")
