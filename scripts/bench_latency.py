import os, sys, time, json
sys.path.insert(0, "/root/repo")
import numpy as np
import cortex_amd
for n in (10_000, 100_000):
    d = 384
    rng = np.random.default_rng(0)
    rows = rng.standard_normal((n, d), dtype=np.float32)
    ids = np.zeros((n, 16), np.uint8); ids[:, 8:] = np.arange(n, dtype=np.uint64).astype(">u8").view(np.uint8).reshape(n, 8)
    h = cortex_amd.HipIndex(d); h.insert_batch(ids, rows)
    q = rows[:64]
    for i in range(200): h.search_arrays(q[i % 64], 5)
    t0 = time.perf_counter()
    for i in range(3000): h.search_arrays(q[i % 64], 5)
    print(json.dumps({"rows": n, "us_per_search": (time.perf_counter() - t0) / 3000 * 1e6}))
