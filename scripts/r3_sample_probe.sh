#!/bin/bash
# how long is batchg's sample pass as a function of the number of sampled tiles?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
for st in 8 32 128 512; do
  CX_BATCHG_SAMPLE_STEP=$st rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sp$st -- python3 $R/scripts/bench_batch_dim.py --rows 1250000 --dim 768 --k 10 --steps 10 > /dev/null 2>&1
  CX_BATCH2=0 CX_BATCHG_SAMPLE_STEP=$st rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sp$st -- python3 $R/scripts/bench_batch_dim.py --rows 1250000 --dim 768 --k 10 --steps 10 > /dev/null 2>&1
  f=$(ls -t $R/gpurun_out/sp$st/*/*kernel_stats.csv | head -1)
  echo "== step $st"; python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'batchg_kernel<0, false' in n or 'dense_topk' in n or 'merge_small' in n or 'cand_select' in n or 'batchg_kernel<0, true' in n:
        print(' ', n.split('(')[0][-40:], r['Calls'], 'avg', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1))
PY
  rm -rf $R/gpurun_out/sp$st
done
