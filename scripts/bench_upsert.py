#!/usr/bin/env python3
"""Startup path of a server: insert every stored embedding (serve.rs:105-123 does it one by one, then rebuild()).
Times cx_upsert_batch from host memory and cx_upsert one at a time."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cortex_amd
n, d = 1_000_000, 768
rng = np.random.default_rng(1)
rows = rng.standard_normal((n, d), dtype=np.float32)
ids = np.zeros((n, 16), np.uint8); ids[:, 8:] = np.arange(n, dtype=np.uint64).astype(">u8").view(np.uint8).reshape(n, 8)
h = cortex_amd.HipIndex(d)
t0 = time.perf_counter(); h.insert_batch(ids, rows); t1 = time.perf_counter() - t0
h2 = cortex_amd.HipIndex(d)
m = 20000
t0 = time.perf_counter()
for i in range(m):
    h2.insert(ids[i].tobytes(), rows[i])
t2 = time.perf_counter() - t0
t0 = time.perf_counter(); h.rebuild(); t3 = time.perf_counter() - t0
print(json.dumps({"rows": n, "dim": d, "upsert_batch_s": t1, "upsert_batch_rows_per_s": n / t1, "upsert_batch_GBs": n * d * 4 / t1 / 1e9,
                  "upsert_single_rows_per_s": m / t2, "rebuild_noop_s": t3}))
