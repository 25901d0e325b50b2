#!/bin/bash
# batch2 after a change: parity tests of the batched paths, then the shapes
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout -k 10 600 python3 -m pytest tests/test_hip_parity.py tests/test_hip_fuzz.py -m gpu -x -q -k "batch or fuzz or lists" 2>&1 | tail -5 || exit 1
cd /tmp; export TMPDIR=/tmp
for a in "1250000 384 10" "1250000 768 10" "5000000 384 10" "1250000 384 5" "1250000 768 32" "10000 384 5"; do set -- $a
  timeout -k 10 200 python3 $R/scripts/bench_batch_dim.py --rows $1 --dim $2 --k $3 --steps 20 2>&1 | tail -1 | cut -c1-260; done
