#!/usr/bin/env python3
"""Where does the post-idle hump of the single-query scan go when the device has just done OTHER work?  After a 1 s idle gap:
(a) 30 scans (the hump: scans 5-40 run ~10 % slow); (b) 70 batched passes (64 queries each, ~20 ms of HBM streaming), then
30 scans; (c) 8 batched passes (~2.5 ms), then 30 scans.  Per-scan kernel time from HIP events."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cortex_amd
from cortex_amd import _lib

L = _lib.load()
n, d, k = 1_000_000, 768, 10
dev = torch.device("cuda", 0)
gen = torch.empty((n, d), dtype=torch.float32, device=dev)
assert L.cx_synth_fill_dev(0, gen.data_ptr(), 20260313, 20260313, 20260315, n // 50, 0, n, d, 1) == 0
ids = np.zeros((n, 16), np.uint8); ids[:, 8:] = np.arange(n, dtype=np.uint64).astype(">u8").view(np.uint8).reshape(n, 8)
ix = cortex_amd.HipIndex(d); ix.reserve(n); ix.insert_batch_dev(ids, gen.data_ptr(), n, d); del gen
q = torch.empty((256, d), dtype=torch.float32, device=dev)
assert L.cx_synth_fill_dev(0, q.data_ptr(), 20260313, 20260314, 20260315, n // 50, 0, 256, d, 0) == 0
o_r = torch.empty((64, k), dtype=torch.int32, device=dev); o_s = torch.empty((64, k), device=dev); o_d = torch.empty((64, k), device=dev)
o_c = torch.empty(64, dtype=torch.int32, device=dev)
ST = torch.cuda.current_stream(dev).cuda_stream
def batches(m):
    for i in range(m): ix.search_batch_dev(q.data_ptr(), 64, k, o_r.data_ptr(), o_s.data_ptr(), o_d.data_ptr(), o_c.data_ptr(), ST)
def scans(m):
    ts = []
    for i in range(m):
        ix.profile_read(reset=True); ix.profile_enable(True)
        ix.search_batch_dev(q.data_ptr() + (i % 256) * d * 4, 1, k, o_r.data_ptr(), o_s.data_ptr(), o_d.data_ptr(), o_c.data_ptr(), ST)
        torch.cuda.synchronize()
        ms, cnt = ix.profile_read(reset=True); ts.append(round(ms * 1e3, 1))
    ix.profile_enable(False)
    return ts
batches(4); scans(80); torch.cuda.synchronize()          # everything built and warm
res = {}
for name, m in (("idle_1s_then_scans", 0), ("idle_1s_70_batched_passes_then_scans", 70), ("idle_1s_8_batched_passes_then_scans", 8), ("idle_1s_then_scans_again", 0)):
    time.sleep(1.0)
    batches(m)
    ts = scans(30)
    res[name] = {"kernel_us": ts, "mean_us": round(sum(ts) / len(ts), 1), "worst_us": max(ts)}
    print(name, ts, flush=True)
print(json.dumps(res))
