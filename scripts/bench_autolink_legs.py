#!/usr/bin/env python3
"""bench.py's all-pairs legs alone (first pass, rescan with existing edges, ordered top-100 lists of every row): one JSON line."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from cortex_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
r = bench.autolink_leg(_lib.load(), 0, 768, True, n=n)
def brief(x):
    return {"wall_ms": round(x["wall_ms"], 3), "edges": x["edges"], "phase_ms": {k: round(v, 3) for k, v in x["phase_ms"].items()},
            "kernel": x["roofline"]["kernel"], "kernel_ms": round(x["roofline"]["avg_kernel_ms"], 3), "frac": round(x["roofline"]["frac"], 4),
            "frac_of_phase": round(x["roofline"]["frac_of_phase"], 4), "clock": round(x["roofline"]["shader_clock_ghz_in_kernel"], 3)}
print(json.dumps({"first": brief(r), "rescan": brief(r["rescan_with_existing_edges"]), "row_list": brief(r["row_list_scan"]), "lists_s": round(r["top100_lists_all_rows"]["seconds"], 4),
                  "full_lists": r["top100_lists_all_rows"]["full_lists"]}))
