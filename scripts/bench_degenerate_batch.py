import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cortex_amd
from cortex_amd import _lib
L = _lib.load()
n, d, k = 1_250_000, 768, 10
dev = torch.device("cuda", 0)
ix = cortex_amd.HipIndex(d); ix.reserve(n)
for lo in range(0, n, 1_000_000):
    m = min(1_000_000, n - lo)
    gen = torch.empty((m, d), dtype=torch.float32, device=dev)
    assert L.cx_synth_fill_dev(0, gen.data_ptr(), 20260313, 20260313, 20260315, n // 50, lo, m, d, 1) == 0
    ids = np.zeros((m, 16), np.uint8); ids[:, 8:] = (np.arange(m, dtype=np.uint64) + lo).astype(">u8").view(np.uint8).reshape(m, 8)
    ix.insert_batch_dev(ids, gen.data_ptr(), m, d); del gen
qd = torch.empty((64, d), dtype=torch.float32, device=dev)
assert L.cx_synth_fill_dev(0, qd.data_ptr(), 20260313, 20260314, 20260315, n // 50, 0, 64, d, 0) == 0
qs = qd.cpu().numpy()
def run(q, tag):
    ix.search_batch_arrays(q, k)
    t0 = time.perf_counter()
    for _ in range(5): r = ix.search_batch_arrays(q, k)
    print(tag, round((time.perf_counter() - t0) / 5 * 1e3, 3), "ms per batch")
    return r
ONLY_FILTER = "--only-filter" in sys.argv   # (under rocprofv3: the kernels of the selective-filter case alone)
r0 = run(qs, "normal batch:")
q1 = qs.copy(); q1[7] = 0.0
if not ONLY_FILTER:
    r1 = run(q1, "one zero query:")
    q2 = qs.copy(); q2[7] = -qs[7]
    r2 = run(q2, "one negated query (few positive cosines?):")
    # check: other queries unchanged, zero query returns rows 0..k-1
    ids0 = [int.from_bytes(bytes(x[8:]), "big") for x in r1[0][7, :k]]
    print("zero query ids", ids0, "scores", r1[1][7, :3])
    assert all(np.array_equal(r0[0][i], r1[0][i]) for i in range(64) if i != 7)
    g = ix.search_arrays(q2[7], k)
    print("negated query batch vs single ids equal:", np.array_equal(r2[0][7, :k], g[0]), r2[1][7, :3], g[1][:3])

# a filter that passes next to nothing (125 of 1.25M rows): no query gets a bound from rows that pass
all_ids = np.zeros((n, 16), np.uint8); all_ids[:, 8:] = np.arange(n, dtype=np.uint64).astype(">u8").view(np.uint8).reshape(n, 8)
kinds = ["common"] * n
for r in range(0, n, 10000): kinds[r] = "rare"
ix.set_metadata_batch(all_ids, kinds, ["kai"] * n)
flt = cortex_amd.VectorFilter(kinds=["rare"])
ix.search_batch_arrays(qs, k, flt)
t0 = time.perf_counter()
for _ in range(50 if ONLY_FILTER else 5): rf = ix.search_batch_arrays(qs, k, flt)
print("selective filter (125 rows pass):", round((time.perf_counter() - t0) / (50 if ONLY_FILTER else 5) * 1e3, 3), "ms per batch")
if ONLY_FILTER: sys.exit(0)
g = ix.search_arrays(qs[3], k, flt)
print("filtered batch vs single ids equal:", np.array_equal(rf[0][3, :k], g[0]), "rows", [int.from_bytes(bytes(x[8:]), "big") for x in g[0][:4]])
flt2 = cortex_amd.VectorFilter(kinds=["common"])
t0 = time.perf_counter()
for _ in range(5): ix.search_batch_arrays(qs, k, flt2)
print("unselective filter (all but 125 pass):", round((time.perf_counter() - t0) / 5 * 1e3, 3), "ms per batch")
