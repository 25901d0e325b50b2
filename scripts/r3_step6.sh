#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3; mkdir -p $O
L=$O/step6.log; : > $L
run() { echo "== $*" >> $L; env "$@" timeout -k 10 200 python3 $R/scripts/bench_autolink.py --reps 8 2>&1 | grep -v amdgpu.ids >> $L; }
for a in 0 1 2 3 4 5 0; do run CX_PAIR_PERSIST=1 CX_PAIR_P_ARM=$a CX_PAIR_P_CLOCK=1; done
python3 $R/scripts/r3_parse.py $L
