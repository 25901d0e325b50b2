#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3; mkdir -p $O
L=$O/step4.log; : > $L
run() { echo "== $*" >> $L; env "$@" timeout -k 10 200 python3 $R/scripts/bench_autolink.py --reps 10 2>&1 | grep -v amdgpu.ids >> $L; }
for round in 1 2; do
  for v in 4 12; do run CX_PAIR_PERSIST=1 CX_PAIR_P_DYN=1 CX_PAIR_P_VAR=$v CX_PAIR_P_CLOCK=1; done
done
python3 $R/scripts/r3_parse.py $L
