#!/bin/bash
# where a batch2 tile's cycles go at 384-d / 768-d (CX_BATCH_DIAG stamps), and the kernel trace of a 384-d step
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/batchdiag; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for d in 384 768; do
  echo "== dim $d plain"; timeout -k 10 200 python3 $R/scripts/bench_batch_dim.py --rows 1250000 --dim $d --k 10 --steps 20 2>&1 | tail -1
  echo "== dim $d diag"; CX_BATCH_DIAG=1 timeout -k 10 200 python3 $R/scripts/bench_batch_dim.py --rows 1250000 --dim $d --k 10 --steps 2 2>&1 | grep -v amdgpu.ids | tail -8
done
echo "== 5M x 384"; timeout -k 10 200 python3 $R/scripts/bench_batch_dim.py --rows 5000000 --dim 384 --k 10 --steps 20 2>&1 | tail -1
echo "== batchg 384"; CX_BATCH2=0 timeout -k 10 200 python3 $R/scripts/bench_batch_dim.py --rows 1250000 --dim 384 --k 10 --steps 20 2>&1 | tail -1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t384 -- python3 $R/scripts/bench_batch_dim.py --rows 1250000 --dim 384 --k 10 --steps 20 > $O/t384.json 2> $O/t384.err
f=$(ls -t $O/t384/*/*kernel_stats.csv | head -1); head -12 $f | cut -c1-200; rm -rf $O/t384
