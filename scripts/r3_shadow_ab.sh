#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout -k 10 900 python3 -m pytest tests/test_hip_autolink.py tests/test_hip_bf16_store.py tests/test_hip_sharded.py tests/test_hip_sharded_abi.py -m gpu -x -q 2>&1 | tail -8 || exit 1
cd /tmp; export TMPDIR=/tmp
timeout -k 10 600 python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config4 > $R/gpurun_out/bench_shadow.json 2> $R/gpurun_out/bench_shadow.err || { tail -5 $R/gpurun_out/bench_shadow.err; exit 1; }
python3 - <<PY
import json
b=json.load(open("$R/gpurun_out/bench_shadow.json"))
e=b["extra"]
for k in ("autolink_allpairs","config5_shard_6.25Mx1024_streaming_ingest"):
    print(k, json.dumps(e.get(k))[:1500])
PY
