#!/bin/bash
# kernel-trace of the batched search at the three shapes VERDICT r2 #4 names: where does a step's time go?
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for a in "6250000 1024 10 bf16" "1250000 384 10 f32" "1250000 768 100 f32" "1250000 384 100 f32"; do
  set -- $a
  tag=${1}x${2}_k${3}_${4}
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bt_$tag -- python3 $R/scripts/bench_batch_dim.py --rows $1 --dim $2 --k $3 --dtype $4 --steps 20 > $O/bt_$tag.json 2> $O/bt_$tag.err || tail -3 $O/bt_$tag.err
  f=$(ls -t $O/bt_$tag/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/bt_${tag}_kernel_stats.csv
  rm -rf $O/bt_$tag
  echo "== $tag"; cat $O/bt_$tag.json | grep -v amdgpu; head -12 $O/bt_${tag}_kernel_stats.csv | cut -d, -f1-5
done
