import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cortex_amd
from oracle import oracle as O
O.build()
def ids_for(n):
    rng = np.random.default_rng(1000); ids = rng.integers(0, 256, size=(n, 16), dtype=np.uint8)
    ids[:, 8:] = np.arange(n, dtype=np.uint64).astype(">u8").view(np.uint8).reshape(n, 8); return ids
d = 768
for n in (3500, 6000):
    rows = O.synth_rows(n, d); ids = ids_for(n)
    h = cortex_amd.HipIndex(d)
    h.insert_batch(ids, rows)
    thr = float(np.float32(0.85))
    for lo in (0, 512, 1003, 2000, 2560, 2992, 3000):
        if lo + 500 > n: continue
        scan = np.arange(lo, lo + 500, dtype=np.uint32)
        res = []
        for p in ("0", "1"):
            os.environ["CX_PAIR_PERSIST"] = p
            ne, ph = h.autolink_pass_timed(100, thr, 50, scan)
            res.append(ne)
        print("n", n, "scan_lo", lo, "edges old/new", res, flush=True)
    h.close()
