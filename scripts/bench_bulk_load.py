#!/usr/bin/env python3
"""Start-up bulk load from stored nodes (serve.rs:105-123): cx_bulk_load_nodes over N bincode `Node` records
(types.rs:26-68) with 768-d embeddings, against the per-node insert loop over the same embeddings."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import cortex_amd
from cortex_amd import _lib
import bincode_ref as B

n, d = int(os.environ.get("N", 500_000)), 768
rng = np.random.default_rng(1)
rows = rng.standard_normal((n, d), dtype=np.float32)
# one template record; id, embedding and created_at are patched per node (same lengths), so building N records is fast
tmpl = B.encode_node(bytes(16), "fact", "A title of typical length for a node", "b" * 400, ["tag-a", "tag-b"], rows[0], "kai",
                     "session-1", None, 0.5, 3, "1970-01-01T00:00:00Z", "2024-01-01T00:00:00Z", "2024-01-01T00:00:00Z", False)
L = len(tmpl)
blob = np.tile(np.frombuffer(tmpl, np.uint8), n).reshape(n, L)
e0 = tmpl.index(rows[0].tobytes())
blob[:, e0:e0 + 4 * d] = rows.view(np.uint8)
blob[:, 8 + 8:8 + 16] = np.arange(n, dtype=np.uint64).astype(">u8").view(np.uint8).reshape(n, 8)
c0 = tmpl.index(b"2024-01-01T00:00:00Z")
secs = rng.integers(0, 86400, n)
hh, mm, ss = secs // 3600, secs // 60 % 60, secs % 60
for off, v in ((11, hh // 10), (12, hh % 10), (14, mm // 10), (15, mm % 10), (17, ss // 10), (18, ss % 10)):
    blob[:, c0 + off] = (v + 48).astype(np.uint8)
offs = (np.arange(n + 1, dtype=np.uint64) * L)
h = cortex_amd.HipIndex(d)
h.reserve(n)
st = _lib.cx_bulk_stats()
Lb = _lib.load()
t0 = time.perf_counter()
rc = Lb.cx_bulk_load_nodes(h._h, n, blob.ctypes.data, offs.ctypes.data, 0, C.byref(st))
t1 = time.perf_counter() - t0
assert rc == 0 and st.indexed == n, (rc, st.indexed)
# newest first
first = h.row_id(0).bytes
assert int.from_bytes(first[8:], "big") == int(np.argmax(secs)) or secs[int.from_bytes(first[8:], "big")] == secs.max()
g = cortex_amd.HipIndex(d)
m = 20000
ids = blob[:m, 8:24].copy()
t0 = time.perf_counter()
for i in range(m):
    g.insert(ids[i].tobytes(), rows[i])
t2 = time.perf_counter() - t0
print(json.dumps({"nodes": n, "dim": d, "record_bytes": L, "bulk_load_s": t1, "nodes_per_s": n / t1, "record_GBs": n * L / t1 / 1e9,
                  "insert_loop_nodes_per_s": m / t2, "host_threads": os.cpu_count()}))
