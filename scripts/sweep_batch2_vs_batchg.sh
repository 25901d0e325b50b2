#!/bin/bash
# The region batch2_kernel still serves (f32 stores at 384 / 768-d below 32,768 rows, or below 131,072 rows for calls of more than
# 256 queries): batch2 (default routing) against batchg forced (CX_BATCH2=0), ms per call.  gpurun, repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { env $1 timeout -k 10 100 python3 $R/scripts/bench_batch.py --rows $2 --dim $3 --nq $4 --k $5 --steps 40 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f' % d['ms_per_batch'])"; }
for dim in 384 768; do
  for spec in "10000 8" "10000 64" "10000 512" "25000 64" "25000 256" "25000 1024" "100000 512" "100000 1024"; do set -- $spec
    for k in 10 100; do
      echo "rows $1 dim $dim nq $2 k $k  batch2 $(run CX_X=0 $1 $dim $2 $k)  batchg $(run CX_BATCH2=0 $1 $dim $2 $k)"
    done
  done
done
