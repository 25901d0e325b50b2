#!/usr/bin/env python3
"""BASELINE configs[4] alone: the streaming-ingest ticks of bench.py's config-5 leg (6.25M x 1024 bf16 shard, batches of 64 / 500)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cortex_amd import _lib
L = _lib.load()
out = bench.config5_leg(L, 0, torch.device("cuda", 0))
free, total = torch.cuda.mem_get_info()
out["device_memory_in_use_GB_after_leg"] = (total - free) / 1e9
print(json.dumps(out))
