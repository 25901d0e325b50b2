#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_hip_autolink.py tests/test_hip_bf16_store.py -x -q -m gpu -k "persistent or ingest or rescan or autolink_pass" > $O/step10_tests.log 2>&1; echo "tests rc=$?" >> $O/step10_tests.log; tail -5 $O/step10_tests.log
timeout -k 10 600 python3 - > $O/step10_config5.log 2>&1 <<'PY'
import json, sys, torch
sys.path.insert(0, ".")
import bench
from cortex_amd import _lib
r = bench.config5_leg(_lib.load(), 0, torch.device("cuda", 0))
print(json.dumps(r))
PY
tail -2 $O/step10_config5.log | cut -c1-3000
echo "== legs" ; timeout -k 10 300 python3 scripts/bench_autolink_legs.py 2>&1 | grep -v amdgpu.ids
