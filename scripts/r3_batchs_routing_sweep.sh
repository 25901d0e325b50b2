#!/bin/bash
# Where the screening pass (batchs.hip) beats batch2 / batchg on small stores: rows x queries x k, both routings (gpurun, repo root).
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rows in 40000 100000; do for dim in 384 768; do for nq in 64 512 1024; do for k in 10 100; do
  for mode in default always never; do
    case $mode in default) e="";; always) e="CX_BATCHS_MIN_ROWS=1024";; never) e="CX_BATCHS=0";; esac
    r=$(env $e timeout -k 10 100 python3 $R/scripts/bench_batch.py --rows $rows --dim $dim --nq $nq --k $k --steps 30 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f' % d['ms_per_batch'])")
    echo "rows $rows dim $dim nq $nq k $k $mode $r ms"
  done
done; done; done; done
