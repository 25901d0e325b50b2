#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3; mkdir -p $O
cd $R
python -m pytest tests/test_hip_autolink.py -x -q -m gpu -k "persistent or autolink_pass_matches or rescan or topk_lists_of_many" > $O/step5_tests.log 2>&1; echo "tests rc=$?" >> $O/step5_tests.log; tail -3 $O/step5_tests.log
L=$O/step5.log; : > $L
run() { echo "== $*" >> $L; env "$@" timeout -k 10 200 python3 $R/scripts/bench_autolink.py --reps 10 2>&1 | grep -v amdgpu.ids >> $L; }
for round in 1 2; do
  run CX_PAIR_PERSIST=1 CX_PAIR_P_CLOCK=1
  run CX_PAIR_PERSIST=1 CX_PAIR_P_ARM=1 CX_PAIR_P_CLOCK=1
  run CX_PAIR_PERSIST=1 CX_PAIR_P_DYN=0 CX_PAIR_P_CLOCK=1
done
run CX_PAIR_PERSIST=1 CX_PAIR_DIAG=1
run CX_PAIR_PERSIST=0
python3 $R/scripts/r3_parse.py $L
