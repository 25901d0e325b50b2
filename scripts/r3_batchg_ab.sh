#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout -k 10 600 python3 -m pytest tests/test_hip_parity.py tests/test_hip_bf16_store.py -m gpu -x -q -k "batch or other_kernel or bf16" 2>&1 | tail -4 || exit 1
cd /tmp; export TMPDIR=/tmp
for a in "6250000 1024 10 bf16" "6250000 1024 10 f32" "1250000 768 100 f32" "1000000 1024 10 f32" "1250000 384 100 f32"; do set -- $a
  timeout -k 10 300 python3 $R/scripts/bench_batch_dim.py --rows $1 --dim $2 --k $3 --dtype $4 --steps 20 2>&1 | tail -1 | cut -c1-300; done
