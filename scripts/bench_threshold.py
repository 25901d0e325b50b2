#!/usr/bin/env python3
"""search_threshold (vector/index.rs:376-388) timing: 1M x 768 f32, thresholds around the dedup / auto-link values."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cortex_amd
from cortex_amd import _lib
L = _lib.load()
n, d = int(os.environ.get("N", 1_000_000)), 768
dev = torch.device("cuda", 0)
gen = torch.empty((n, d), dtype=torch.float32, device=dev)
assert L.cx_synth_fill_dev(0, gen.data_ptr(), 20260313, 20260313, 20260315, max(1, n // 50), 0, n, d, 1) == 0
ids = np.zeros((n, 16), np.uint8); ids[:, 8:] = np.arange(n, dtype=np.uint64).astype(">u8").view(np.uint8).reshape(n, 8)
h = cortex_amd.HipIndex(d); h.insert_batch_dev(ids, gen.data_ptr(), n, d)
q = gen[:64].cpu().numpy(); del gen
out = {}
for thr in (0.92, 0.85, 0.75, 0.0):
    for _ in range(2): h.search_threshold(q[0], thr)
    t0 = time.perf_counter(); tot = 0
    reps = 20 if thr > 0 else 3
    for i in range(reps): tot += len(h.search_threshold(q[i % 64], thr))
    out[str(thr)] = {"ms": (time.perf_counter() - t0) / reps * 1e3, "avg_results": tot / reps}
t0 = time.perf_counter()
for i in range(20): h.search(q[i % 64], 10)
out["search_k10_ms"] = (time.perf_counter() - t0) / 20 * 1e3
print(json.dumps(out))
