#!/bin/bash
# Collects the round's judged artifacts on the GPU box (run through gpurun from the repo root):
#   bench.py JSON line, rocprofv3 kernel-trace summaries for the three workloads, and the HBM PMC passes of the
#   dominant kernel (FETCH_SIZE and WRITE_SIZE in separate runs, as MI355X_MICROARCH.md prescribes).
#   scripts/profile_round.sh [bench|batch1024|knn|autolink|all] — in parts, each within one gpurun call's limit;
#   scripts/profile_round_b.sh: the batched search's shapes and counters.
# Output: gpurun_out/prof_round/ ; copy the *_kernel_stats.csv / *.json you want judged into profiles/rNN/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_round
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
step() { echo "== $1"; }
PART=${1:-all}
want() { [ "$PART" = all ] || [ "$PART" = "$1" ]; }
if want bench; then
step bench; timeout -k 10 900 python3 $R/bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
# the driver's own command line (round 1: 0.77 on the post-idle clock transient; the pre-roll carries the device past it)
step bench-driver-cmdline; timeout -k 10 300 python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-autolink --no-config4 > $O/bench_steps20_warmup5.json 2> $O/bench_steps20.err || { tail -5 $O/bench_steps20.err; exit 1; }
step cold-probe; timeout -k 10 200 python3 $R/scripts/cold_probe.py > $O/cold_probe.log 2>&1 && cp $R/gpurun_out/cold_probe.json $O/cold_start_probe.json
step legs; timeout -k 10 300 python3 $R/scripts/bench_autolink_legs.py > $O/autolink_legs.json 2>/dev/null
step lists; timeout -k 10 200 python3 $R/scripts/bench_lists.py 100000 768 > $O/top100_lists_100kx768.log 2>&1
fi
if want batch1024; then
step batch1024-trace; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b1024_trace -- python3 $R/scripts/bench_batch_dim.py --rows 1000000 --dim 1024 > $O/batch64_1Mx1024.json 2> $O/b1024_trace.err || { tail -5 $O/b1024_trace.err; exit 1; }
step batch1024-fetch; timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/b1024_fetch -- python3 $R/scripts/bench_batch_dim.py --rows 1000000 --dim 1024 --steps 5 > $O/b1024_fetch.json 2> $O/b1024_fetch.err || { tail -5 $O/b1024_fetch.err; exit 1; }
step batch1024-write; timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/b1024_write -- python3 $R/scripts/bench_batch_dim.py --rows 1000000 --dim 1024 --steps 5 > $O/b1024_write.json 2> $O/b1024_write.err || { tail -5 $O/b1024_write.err; exit 1; }
step batch-shapes; for a in "4000000 1024 10 f32" "1000000 1024 100 f32" "1000000 1024 256 f32" "1250000 768 100 f32" "1250000 384 100 f32" "2000000 512 10 f32" "1000000 1536 10 f32" "1000000 1024 10 bf16" "6250000 1024 10 bf16" "1250000 768 100 bf16"; do set -- $a; timeout -k 10 200 python3 $R/scripts/bench_batch_dim.py --rows $1 --dim $2 --k $3 --dtype $4 --steps 10 2>/dev/null; done > $O/batch64_other_shapes.jsonl
step single-bf16; for a in "1000000 768" "1000000 1024" "1000000 384"; do set -- $a; timeout -k 10 200 python3 $R/scripts/bench_batch_dim.py --rows $1 --dim $2 --single --steps 200 --dtype bf16 2>/dev/null; done > $O/single_query_bf16_store.jsonl
# the stand-alone probes behind profiles/rNN/tuning.md section 4 (built here if the snapshot has no binaries)
step probes; P=$R/scripts/probes
[ -x $P/_shape_probe ] || (cd $P && hipcc --offload-arch=gfx950 -O3 -std=c++17 -o _shape_probe shape_probe.hip 2>/dev/null)
[ -x $P/_mfma_shape_probe ] || (cd $P && hipcc --offload-arch=gfx950 -O3 -std=c++17 -o _mfma_shape_probe mfma_shape_probe.hip 2>/dev/null)
timeout -k 10 120 $P/_shape_probe 1 40 > $O/read_shape_probe.log 2>&1; timeout -k 10 120 $P/_mfma_shape_probe > $O/mfma_shape_probe.log 2>&1
fi
if want knn; then
step knn-trace; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/knn_trace -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-autolink --no-config4 > $O/knn_trace.json 2> $O/knn_trace.err || { tail -5 $O/knn_trace.err; exit 1; }
step knn-fetch; timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/knn_fetch -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-autolink --no-config4 > $O/knn_fetch.json 2> $O/knn_fetch.err || { tail -5 $O/knn_fetch.err; exit 1; }
step knn-write; timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/knn_write -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-autolink --no-config4 > $O/knn_write.json 2> $O/knn_write.err || { tail -5 $O/knn_write.err; exit 1; }
fi
if want autolink; then
step autolink-trace; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/al_trace -- python3 $R/scripts/bench_autolink.py --reps 24 > $O/autolink.json 2> $O/al_trace.err || { tail -5 $O/al_trace.err; exit 1; }
# MFMA pipe utilisation of the all-pairs filter GEMM (north_star: ">= 50 % MFMA utilisation"): counters only, program directly after `--`
step autolink-mfma; timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/al_mfma -- python3 $R/scripts/bench_autolink.py --reps 24 > $O/al_mfma.json 2> $O/al_mfma.err || { tail -5 $O/al_mfma.err; exit 1; }
fi
# keep only the summaries (traces are large)
for d in knn_trace al_trace b1024_trace; do f=$(ls -t $O/$d/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
for d in knn_fetch knn_write al_mfma b1024_fetch b1024_write; do f=$(ls -t $O/$d/*/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python3 - "$f" > $O/${d}_summary.json <<'PY'
import csv, sys, json, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"].split("(")[0][:80], r["Counter_Name"])
    agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
print(json.dumps([{"kernel": k[0], "counter": k[1], "sum": v[0], "dispatches": v[1], "per_dispatch": v[0] / v[1]} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:8]], indent=1))
PY
done
# MFMA utilisation of pair_filter_p_kernel: busy cycles (summed over all SIMDs) / (kernel cycles x 256 CUs x 4 SIMDs);
# GRBM_GUI_ACTIVE is reported per XCD and summed over the 8 of them
[ -f $O/al_mfma_summary.json ] && python3 - $O/al_mfma_summary.json > $O/autolink_mfma_utilisation.json <<'PY'
import json, sys
rows = json.load(open(sys.argv[1]))
def per(kern, ctr):
    for r in rows:
        if kern in r["kernel"] and r["counter"] == ctr:
            return r["per_dispatch"], r["dispatches"]
    return None, 0
busy, nb = per("pair_filter_p_kernel", "SQ_VALU_MFMA_BUSY_CYCLES")
act, na = per("pair_filter_p_kernel", "GRBM_GUI_ACTIVE")
out = {"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -- python3 scripts/bench_autolink.py --reps 24",
       "kernel": "cx::pair_filter_p_kernel", "dispatches": nb,
       "per_launch": {"SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE_sum_over_8_XCDs": act}}
if busy and act:
    cyc = act / 8.0
    out["derived"] = {"kernel_cycles": cyc, "mfma_pipe_utilisation": busy / (cyc * 256 * 4)}
print(json.dumps(out, indent=1))
PY
rm -rf $O/knn_trace $O/al_trace $O/batch_trace $O/b1024_trace $O/knn_fetch $O/knn_write $O/batch_fetch $O/al_mfma $O/b1024_fetch $O/b1024_write
ls $O
