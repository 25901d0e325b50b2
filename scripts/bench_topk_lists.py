#!/usr/bin/env python3
"""The auto-linker's ordered top-100 lists for every row (cx_topk_lists_rows; SURVEY a14': what the reference's
default configuration — legacy structural rules on — needs from the engine): N x dim corpus, all rows scanned."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cortex_amd
from cortex_amd import _lib
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=100_000)
ap.add_argument("--dim", type=int, default=384)
ap.add_argument("--k", type=int, default=100)
a = ap.parse_args()
L = _lib.load()
n, d = a.rows, a.dim
gen = torch.empty((n, d), dtype=torch.float32, device="cuda:0")
assert L.cx_synth_fill_dev(0, gen.data_ptr(), 20260313, 20260313, 20260315, max(1, n // 50), 0, n, d, 1) == 0
ids = np.zeros((n, 16), np.uint8); ids[:, 8:] = np.arange(n, dtype=np.uint64).astype(">u8").view(np.uint8).reshape(n, 8)
h = cortex_amd.HipIndex(d); h.insert_batch_dev(ids, gen.data_ptr(), n, d); del gen
h.topk_lists_rows(a.k, np.arange(2048, dtype=np.uint32))
t0 = time.perf_counter(); r, s, c = h.topk_lists_rows(a.k, None); t = time.perf_counter() - t0
print(json.dumps({"rows": n, "dim": d, "k": a.k, "seconds": t, "lists_per_s": n / t, "full_lists": int((c == a.k).sum())}))
