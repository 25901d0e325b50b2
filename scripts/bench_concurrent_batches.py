#!/usr/bin/env python3
"""Concurrent readers (the trait's &self contract: RwLock read side): T host threads, each its own stream of batched searches
over the same index — how do several batched-search passes share the device?  Prints per-thread ms per batch beside the
single-thread figure."""
import argparse, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cortex_amd
from cortex_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_250_000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--steps", type=int, default=40)
a = ap.parse_args()
L = _lib.load()
n, d, k = a.rows, a.dim, a.k
dev = torch.device("cuda", 0)
ix = cortex_amd.HipIndex(d); ix.reserve(n)
for lo in range(0, n, 1_000_000):
    m = min(1_000_000, n - lo)
    gen = torch.empty((m, d), dtype=torch.float32, device=dev)
    assert L.cx_synth_fill_dev(0, gen.data_ptr(), 20260313, 20260313, 20260315, n // 50, lo, m, d, 1) == 0
    ids = np.zeros((m, 16), np.uint8); ids[:, 8:] = (np.arange(m, dtype=np.uint64) + lo).astype(">u8").view(np.uint8).reshape(m, 8)
    ix.insert_batch_dev(ids, gen.data_ptr(), m, d); del gen
qd = torch.empty((256, d), dtype=torch.float32, device=dev)
assert L.cx_synth_fill_dev(0, qd.data_ptr(), 20260313, 20260314, 20260315, n // 50, 0, 256, d, 0) == 0
qs = qd.cpu().numpy()
ref = ix.search_batch_arrays(qs[:64], k)
out = {}
for T in (1, 2, 4):
    res = [None] * T
    def work(t):
        t0 = time.perf_counter()
        ok = True
        for i in range(a.steps):
            got = ix.search_batch_arrays(qs[:64], k)
            ok = ok and all(np.array_equal(x, y) for x, y in zip(ref, got))
        res[t] = ((time.perf_counter() - t0) / a.steps * 1e3, ok)
    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    wall = time.perf_counter() - t0
    out[f"{T}_threads"] = {"ms_per_batch_per_thread": [round(r[0], 3) for r in res], "identical_results": all(r[1] for r in res),
                           "batches_per_s_total": T * a.steps / wall}
print(json.dumps({"rows": n, "dim": d, "k": k, **out}))
