#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_hip_autolink.py tests/test_hip_sharded_abi.py tests/test_hip_bf16_store.py -x -q -m gpu > $O/step9_tests.log 2>&1; echo "tests rc=$?" >> $O/step9_tests.log; tail -5 $O/step9_tests.log
timeout -k 10 120 python3 scripts/fuzz_autolink.py --seconds 45 > $O/step9_fuzz.log 2>&1; echo "fuzz rc=$?" >> $O/step9_fuzz.log; tail -3 $O/step9_fuzz.log
L=$O/step9.log; : > $L
for arm in "CX_PAIR_PERSIST=1" "CX_PAIR_PERSIST=0"; do
  echo "== $arm" >> $L; env $arm timeout -k 10 300 python3 scripts/bench_autolink_legs.py 2>&1 | grep -v amdgpu.ids >> $L
done
cat $L
