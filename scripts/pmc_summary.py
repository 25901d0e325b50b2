#!/usr/bin/env python3
"""Sum a rocprofv3 counter_collection.csv per (kernel, counter): python3 scripts/pmc_summary.py <dir-or-csv> [kernel-substring]"""
import collections, csv, glob, json, os, sys

p = sys.argv[1]
if os.path.isdir(p):
    fs = sorted(glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    p = fs[-1] if fs else None
if not p:
    print("[]"); sys.exit(0)
want = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(p)):
    name = r["Kernel_Name"].split("(")[0][:90]
    if want and want not in name:
        continue
    k = (name, r["Counter_Name"])
    agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
print(json.dumps([{"kernel": k[0], "counter": k[1], "dispatches": v[1], "per_dispatch": v[0] / v[1]}
                  for k, v in sorted(agg.items(), key=lambda kv: (kv[0][0], kv[0][1]))], indent=1))
