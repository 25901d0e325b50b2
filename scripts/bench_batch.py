#!/usr/bin/env python3
"""Batched search timing (BASELINE config 4's per-GPU inner loop): n x 768 f32 shard, B=64, k=10."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import cortex_amd
from cortex_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_250_000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--nq", type=int, default=64)
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--steps", type=int, default=30)
a = ap.parse_args()
L = _lib.load()
n, d, nq, k = a.rows, a.dim, a.nq, a.k
dev = torch.device("cuda", 0)
gen = torch.empty((n, d), dtype=torch.float32, device=dev)
assert L.cx_synth_fill_dev(0, gen.data_ptr(), 20260313, 20260313, 20260315, max(1, n // 50), 0, n, d, 1) == 0
ids = np.zeros((n, 16), np.uint8); ids[:, 8:] = np.arange(n, dtype=np.uint64).astype(">u8").view(np.uint8).reshape(n, 8)
h = cortex_amd.HipIndex(d); h.insert_batch_dev(ids, gen.data_ptr(), n, d); del gen
q = torch.empty((nq, d), dtype=torch.float32, device=dev)
assert L.cx_synth_fill_dev(0, q.data_ptr(), 20260313, 20260314, 20260315, max(1, n // 50), 0, nq, d, 0) == 0
rows = torch.zeros((nq, k), dtype=torch.int32, device=dev); sc = torch.zeros((nq, k), device=dev); di = torch.zeros((nq, k), device=dev)
cnt = torch.zeros(nq, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
def step(): h.search_batch_dev(q.data_ptr(), nq, k, rows.data_ptr(), sc.data_ptr(), di.data_ptr(), cnt.data_ptr(), st)
for _ in range(3): step()
torch.cuda.synchronize(); h.profile_read(True); h.profile_enable(True)
t0 = time.perf_counter()
for _ in range(a.steps): step()
torch.cuda.synchronize(); t = time.perf_counter() - t0
ms, cntk = h.profile_read(True)
screened = n >= 131072 and d % 128 == 0 and d <= 1024 and nq >= 3   # batchs.hip streams the 2-byte screening copy, not the f32 rows
bytes_ = n * d * (2.0 if screened else 4.0)
print(json.dumps({"rows": n, "dim": d, "nq": nq, "k": k, "ms_per_batch": t / a.steps * 1e3, "queries_per_s": nq * a.steps / t,
                  "kernel_ms": ms / max(1, cntk), "bytes_per_launch": bytes_, "hbm_GBs": bytes_ / (ms / max(1, cntk) * 1e-3) / 1e9,
                  "frac_of_8TBs": bytes_ / (ms / max(1, cntk) * 1e-3) / 8e12, "frac_step_of_8TBs": bytes_ / (t / a.steps) / 8e12,
                  "store_equivalent_step_GBs": n * d * 4.0 / (t / a.steps) / 1e9}))
