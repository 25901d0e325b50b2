"""A stream of batches on 0 (one stream), 2 and 4 rotating HIP streams (ShardedKnn.submit), per batch shape: which depth a rank of the
sharded search should run.  gpurun, repo root:  python3 scripts/probe_stream_depth.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cortex_amd import _lib
L = _lib.load()
dev = torch.device("cuda", 0)
for d, B, k in ((384, 128, 10), (384, 64, 10), (768, 64, 10), (512, 128, 10), (384, 128, 100)):
    for depth in ("0", "2", "4"):
        os.environ["CX_SHARDED_STREAMS"] = depth
        r = bench.config4_leg(L, 0, dev, d=d, B=B, k=k, steps=100, n=1_250_000 if d != 512 else 1_000_000)
        print(json.dumps({"dim": d, "B": B, "k": k, "streams": int(depth), "one_stream_ms": round(r["ms_per_step"], 4),
                          "stream_of_batches_ms": round(r["stream_of_batches"]["ms_per_step"], 4)}), flush=True)
