#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/full_tests.log 2>&1; echo "tests rc=$?" >> $O/full_tests.log; tail -6 $O/full_tests.log
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "bench rc=$?"; tail -3 $O/bench_steps20.err
python3 - <<'PY' $O/bench_steps20.json
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", round(r["value"], 1), "ms/step", round(r["ms_per_step"], 4), "frac", round(r["roofline"]["frac"], 4))
e = r["extra"]
print("steady", round(e["steady_state"]["value"], 1), round(e["steady_state"]["roofline"]["frac"], 4))
print("cold_burst", e["cold_burst"]["mean_kernel_ms"], e["cold_burst"]["frac_of_hbm_peak_mean"])
for k in ("autolink_allpairs", "autolink_allpairs_bench_corpus"):
    x = e[k]; print(k, "wall", round(x["wall_ms"], 2), "kernel", x["roofline"]["kernel"], round(x["roofline"]["avg_kernel_ms"], 3), "frac", round(x["roofline"]["frac"], 4), "phase frac", round(x["roofline"]["frac_of_phase"], 4))
x = e["autolink_allpairs"]["rescan_with_existing_edges"]; print("rescan wall", round(x["wall_ms"], 2), "frac", round(x["roofline"]["frac"], 4))
print("lists", e["autolink_allpairs"]["top100_lists_all_rows"]["seconds"])
print("recall", e.get("recall_at_k_vs_exact"), e.get("recall_at_k_vs_exact_search_batch"), e.get("single_vs_batch_self_check"))
for k in e:
    if k.startswith("config4") or k.startswith("config5") or k.startswith("config2"):
        x = e[k]
        if "roofline" in x: print(k, round(x.get("queries_per_s", 0), 1), round(x.get("ms_per_step", 0), 3), round(x["roofline"]["frac"], 4))
        else: print(k, {kk: (vv if not isinstance(vv, dict) else {a: b for a, b in vv.items() if a in ("tick_ms", "filter_ms", "upsert_ms")}) for kk, vv in x.items() if kk.startswith("batch_")})
print("cpu", r.get("cpu_baseline"))
PY
