#!/bin/bash
# The batched-search artifacts of a round (batchs.hip), run through gpurun from the repo root; bench.py itself and the other
# workloads' steps are scripts/profile_round.sh's.  Output: gpurun_out/prof_round/ ; scripts/collect_profiles.py copies the judged
# summaries into profiles/rNN/.  Counters are collected in their own passes (FETCH_SIZE, WRITE_SIZE), program directly
# after `--`.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_round
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
step() { echo "== $1"; }
if [ "${1:-all}" != "pmc" ]; then
step batch-shapes; for a in "1250000 768 10 f32" "1250000 768 100 f32" "1250000 384 10 f32" "1250000 384 100 f32" "5000000 768 10 f32" "5000000 384 10 f32" "4000000 1024 10 f32" "1000000 1024 100 f32" "1000000 1024 256 f32" "2000000 512 10 f32" "1000000 1024 10 bf16" "6250000 1024 10 bf16" "6250000 1024 100 bf16" "1250000 768 100 bf16" "150000 384 10 f32" "300000 768 10 f32"; do set -- $a; timeout -k 10 200 python3 $R/scripts/bench_batch_dim.py --rows $1 --dim $2 --k $3 --dtype $4 --steps 40 2>/dev/null; done > $O/batch64_other_shapes.jsonl
fi
if [ "${1:-all}" != "bench" ]; then
step batch-trace; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/batch_trace -- python3 $R/scripts/bench_batch.py > $O/batch.json 2> $O/batch_trace.err || { tail -5 $O/batch_trace.err; exit 1; }
step batch-fetch; timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/batch_fetch -- python3 $R/scripts/bench_batch.py --steps 5 > $O/batch_fetch.json 2> $O/batch_fetch.err || { tail -5 $O/batch_fetch.err; exit 1; }
step batch-write; timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/batch_write -- python3 $R/scripts/bench_batch.py --steps 5 > $O/batch_write.json 2> $O/batch_write.err || { tail -5 $O/batch_write.err; exit 1; }
step shard1024-trace; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b1024_trace -- python3 $R/scripts/bench_batch_dim.py --rows 6250000 --dim 1024 --dtype bf16 > $O/batch64_6.25Mx1024_bf16.json 2> $O/b1024_trace.err || { tail -5 $O/b1024_trace.err; exit 1; }
step shard1024-fetch; timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/b1024_fetch -- python3 $R/scripts/bench_batch_dim.py --rows 6250000 --dim 1024 --dtype bf16 --steps 5 > $O/b1024_fetch.json 2> $O/b1024_fetch.err || { tail -5 $O/b1024_fetch.err; exit 1; }
step shard1024-write; timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/b1024_write -- python3 $R/scripts/bench_batch_dim.py --rows 6250000 --dim 1024 --dtype bf16 --steps 5 > $O/b1024_write.json 2> $O/b1024_write.err || { tail -5 $O/b1024_write.err; exit 1; }
for d in batch_trace b1024_trace; do f=$(ls -t $O/$d/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
for d in batch_fetch batch_write b1024_fetch b1024_write; do f=$(ls -t $O/$d/*/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python3 - "$f" > $O/${d}_summary.json <<'PY'
import csv, sys, json, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"].split("(")[0][:80], r["Counter_Name"])
    agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
print(json.dumps([{"kernel": k[0], "counter": k[1], "sum": v[0], "dispatches": v[1], "per_dispatch": v[0] / v[1]} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:8]], indent=1))
PY
done
rm -rf $O/batch_trace $O/b1024_trace $O/batch_fetch $O/batch_write $O/b1024_fetch $O/b1024_write
fi
ls $O
