#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_autolink.py -x -q -m gpu -k "persistent or autolink_pass_matches or rescan or topk_lists_of_many" > $O/step7_tests.log 2>&1; echo "tests rc=$?" >> $O/step7_tests.log; tail -3 $O/step7_tests.log
L=$O/step7.log; : > $L
run() { echo "== $*" >> $L; env "$@" timeout -k 10 200 python3 $R/scripts/bench_autolink.py --reps 8 2>&1 | grep -v amdgpu.ids >> $L; }
for round in 1 2; do
  run CX_PAIR_P_BM=256 CX_PAIR_P_CLOCK=1
  run CX_PAIR_P_BM=128 CX_PAIR_P_CLOCK=1
done
run CX_PAIR_P_BM=128 CX_PAIR_P_ARM=4 CX_PAIR_P_CLOCK=1
run CX_PAIR_P_BM=128 CX_PAIR_P_ARM=2 CX_PAIR_P_CLOCK=1
run CX_PAIR_P_BM=128 CX_PAIR_DIAG=1
run CX_PAIR_PERSIST=0
python3 $R/scripts/r3_parse.py $L | grep -v "diag\] 6"
grep "diag\]" $L | tail -1
