#!/usr/bin/env python3
"""Why do the first ~25 scans after start-up run 13 % slow?  Per-launch HIP-event times of
the scan kernel from the very first launch on, for (a) a freshly loaded index, (b) the same index after an idle
gap, (c) after a pass of a plain read over the store (TLB / page-table warm) — to tell clock ramp from first touch."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import cortex_amd
from cortex_amd import _lib

L = _lib.load()
n, d, k = 1_000_000, 768, 10
dev = torch.device("cuda", 0)
gen = torch.empty((n, d), dtype=torch.float32, device=dev)
assert L.cx_synth_fill_dev(0, gen.data_ptr(), 20260313, 20260313, 20260315, n // 50, 0, n, d, 1) == 0
ids = np.zeros((n, 16), np.uint8)
ids[:, 8:] = np.arange(n, dtype=np.uint64).astype(">u8").view(np.uint8).reshape(n, 8)
ix = cortex_amd.HipIndex(d)
ix.reserve(n)
ix.insert_batch_dev(ids, gen.data_ptr(), n, d)
q = torch.empty((256, d), dtype=torch.float32, device=dev)
assert L.cx_synth_fill_dev(0, q.data_ptr(), 20260313, 20260314, 20260315, n // 50, 0, 256, d, 0) == 0
out_r = torch.empty(k, dtype=torch.int32, device=dev)
out_s = torch.empty(k, dtype=torch.float32, device=dev)
out_d = torch.empty(k, dtype=torch.float32, device=dev)
out_c = torch.empty(1, dtype=torch.int32, device=dev)
STREAM = torch.cuda.current_stream(dev).cuda_stream
torch.cuda.synchronize()


def burst(m, label):
    ts = []
    for i in range(m):
        ix.profile_read(reset=True)
        ix.profile_enable(True)
        ix.search_batch_dev(q.data_ptr() + (i % 256) * d * 4, 1, k, out_r.data_ptr(), out_s.data_ptr(), out_d.data_ptr(), out_c.data_ptr(), STREAM)
        torch.cuda.synchronize()
        ms, cnt = ix.profile_read(reset=True)
        ts.append(ms)
    ix.profile_enable(False)
    print(label, " ".join(f"{t*1e3:.0f}" for t in ts), flush=True)
    return ts


res = {}
res["cold"] = burst(60, "cold     ")
res["warm"] = burst(20, "warm     ")
time.sleep(2.0)
res["idle2s"] = burst(20, "idle 2s  ")
time.sleep(0.2)
res["idle0.2s"] = burst(20, "idle 0.2s")
# back-to-back without a host sync between launches (bench.py's loop shape)
ix.profile_read(reset=True)
ix.profile_enable(True)
for i in range(60):
    ix.search_batch_dev(q.data_ptr() + (i % 256) * d * 4, 1, k, out_r.data_ptr(), out_s.data_ptr(), out_d.data_ptr(), out_c.data_ptr(), STREAM)
torch.cuda.synchronize()
ms, cnt = ix.profile_read(reset=True)
print(f"back-to-back 60: avg {ms / cnt * 1e3:.0f} us", flush=True)
res["b2b_avg_us"] = ms / cnt * 1e3

# Round 3, one remedy tried once: does a trickle of work during the idle gap keep the board where it
# was?  (a) a 256 MiB read every 5 ms (about 1 % duty) from a side stream while the caller idles for 1 s; (b) the same gap with
# eight microsecond-scale dummy launches right before the burst; (c) the plain 1 s gap again, for reference on this board.
import threading
pad = torch.empty(64 << 20, dtype=torch.float32, device=dev)      # 256 MiB
side = torch.cuda.Stream(device=dev)


def idle_with_trickle(seconds, period):
    stop = time.perf_counter() + seconds
    with torch.cuda.stream(side):
        while time.perf_counter() < stop:
            pad.sum()
            side.synchronize()
            time.sleep(period)


time.sleep(1.0)
res["idle1s_again"] = burst(40, "idle 1s  ")
idle_with_trickle(1.0, 0.005)
res["idle1s_trickle_5ms"] = burst(40, "trickle 5")
idle_with_trickle(1.0, 0.050)
res["idle1s_trickle_50ms"] = burst(40, "trickle50")
time.sleep(1.0)
tiny = torch.empty(1024, dtype=torch.float32, device=dev)
for _ in range(8):
    tiny.add_(1.0)
res["idle1s_then_8_dummy_launches"] = burst(40, "dummies  ")
_out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
os.makedirs(_out, exist_ok=True)
json.dump(res, open(os.path.join(_out, "cold_probe.json"), "w"))
