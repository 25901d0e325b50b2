#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3; mkdir -p $O
cd $R
L=$O/step8.log; : > $L
for arm in "CX_PAIR_PERSIST=1" "CX_PAIR_PERSIST=0" "CX_PAIR_PERSIST=1" "CX_PAIR_PERSIST=0"; do
  echo "== $arm" >> $L; env $arm timeout -k 10 300 python3 scripts/bench_autolink_legs.py 2>&1 | grep -v amdgpu.ids >> $L
done
cat $L
timeout -k 10 900 python -m pytest tests/test_hip_autolink.py tests/test_hip_bf16_store.py tests/test_hip_sharded_abi.py -x -q -m gpu > $O/step8_tests.log 2>&1; echo "tests rc=$?" >> $O/step8_tests.log; tail -4 $O/step8_tests.log
