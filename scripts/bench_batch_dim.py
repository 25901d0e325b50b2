#!/usr/bin/env python3
"""Batched search at a given row width: 64 queries per step over n rows, k = 10 (cx_search_batch_dev), HIP events around
the scan kernel — the config-4 measurement at other dims (1024 = BGE-large, BASELINE config 5's width)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cortex_amd
from cortex_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_000_000)
ap.add_argument("--dim", type=int, default=1024)
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--dtype", default="f32", help="storage dtype of the index: f32 | bf16 (cx_create_ex)")
ap.add_argument("--single", action="store_true", help="time single-query scans instead of batches")
a = ap.parse_args()
L = _lib.load()
n, d, k, B = a.rows, a.dim, a.k, a.batch
dev = torch.device("cuda", 0)
ix = cortex_amd.HipIndex(d, dtype=a.dtype); ix.reserve(n)
for lo in range(0, n, 1_000_000):
    m = min(1_000_000, n - lo)
    gen = torch.empty((m, d), dtype=torch.float32, device=dev)
    assert L.cx_synth_fill_dev(0, gen.data_ptr(), 20260313, 20260313, 20260315, n // 50, lo, m, d, 1) == 0
    ids = np.zeros((m, 16), np.uint8); ids[:, 8:] = (np.arange(m, dtype=np.uint64) + lo).astype(">u8").view(np.uint8).reshape(m, 8)
    ix.insert_batch_dev(ids, gen.data_ptr(), m, d); del gen
qs = torch.empty((256, d), dtype=torch.float32, device=dev)
assert L.cx_synth_fill_dev(0, qs.data_ptr(), 20260313, 20260314, 20260315, n // 50, 0, 256, d, 0) == 0
o_rows = torch.empty((B, k), dtype=torch.int32, device=dev); o_sc = torch.empty((B, k), dtype=torch.float32, device=dev)
o_di = torch.empty((B, k), dtype=torch.float32, device=dev); o_cnt = torch.empty(B, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
if a.single:
    B = 1
one = lambda i: ix.search_batch_dev(qs.data_ptr() + ((i * B) % (256 - B + 1)) * d * 4, B, k, o_rows.data_ptr(), o_sc.data_ptr(), o_di.data_ptr(), o_cnt.data_ptr(), stream)
for i in range(8): one(i)
torch.cuda.synchronize()
ix.profile_read(reset=True); ix.profile_enable(True)
t0 = time.perf_counter()
for i in range(a.steps): one(i)
torch.cuda.synchronize()
el = time.perf_counter() - t0
ix.profile_enable(False)
ms, cnt = ix.profile_read(reset=True)
avg = ms / max(1, cnt); store = float(n) * d * (2 if a.dtype == "bf16" else 4)
screened = (not a.single) and n >= 131072 and d % 128 == 0 and d <= 1024 and B >= 3   # batchs.hip streams the 2-byte screening copy
algo = float(n) * d * 2 if screened else store
print(json.dumps({"rows": n, "dim": d, "dtype": a.dtype, "k": k, "batch": B, "queries_per_s": a.steps * B / el, "ms_per_step": el / a.steps * 1e3,
                  "kernel_ms": avg, "launches": cnt, "hbm_GBs": algo / (avg * 1e-3) / 1e9 if avg else None,
                  "frac_of_8TBs": algo / (avg * 1e-3) / 8e12 if avg else None, "bytes_per_launch": algo,
                  "frac_step_of_8TBs": algo / (el / a.steps) / 8e12, "store_equivalent_step_GBs": store / (el / a.steps) / 1e9}))
