import sys, json, os
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
import bench
from cortex_amd import _lib
L = _lib.load()
dev = torch.device("cuda", 0)
r = bench.config4_sharded_leg(L, 0, dev, 0, 1, total=1_250_000, steps=200)
print(json.dumps({"streams": os.environ.get("CX_SHARDED_STREAMS", "4"), "ms_per_step": r["ms_per_step"], "queries_per_s": r["queries_per_s"], "kernel_ms": r["roofline"]["avg_kernel_ms"]}))
