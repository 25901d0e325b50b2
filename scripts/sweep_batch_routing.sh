#!/bin/bash
# Which kernel family serves a batched search fastest: rows x dim x queries x k, the screening pass (batchs.hip) forced against
# batch2 / batchg forced (gpurun, repo root; the switches are read once per process).  ms per call.
#   scripts/sweep_batch_routing.sh "<rows list>" "<dim list>" "<nq list>" "<k list>"
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROWS=${1:-"2000 10000 25000"}; DIMS=${2:-"384 1024"}; NQS=${3:-"8 64 256"}; KS=${4:-"10 100"}
for rows in $ROWS; do for dim in $DIMS; do for nq in $NQS; do for k in $KS; do
  line="rows $rows dim $dim nq $nq k $k"
  for mode in screening other; do
    case $mode in screening) e="CX_BATCHS_MIN_ROWS=1 CX_BATCHS_SMALL_ROWS=1";; other) e="CX_BATCHS=0";; esac
    r=$(env $e timeout -k 10 100 python3 $R/scripts/bench_batch.py --rows $rows --dim $dim --nq $nq --k $k --steps 40 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f' % d['ms_per_batch'])")
    line="$line  $mode $r"
  done
  echo "$line"
done; done; done; done
