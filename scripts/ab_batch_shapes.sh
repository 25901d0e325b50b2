#!/bin/bash
# Same-box A/B of the batched search: this tree's library against cortex_amd/lib/libcortex_hip_base.so (scripts/build_base_lib.sh <rev>),
# (LIBS="new base other": several), alternating, two rounds; shapes "rows dim k" as arguments.  gpurun, repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
  for shape in "$@"; do read -r a1 a2 a3 <<< "$shape"
    for lib in ${LIBS:-new base}; do
      if [ $lib != new ]; then e="CORTEX_HIP_LIB=$R/cortex_amd/lib/libcortex_hip_$lib.so"; else e="CX_X=0"; fi
      echo -n "rows $a1 dim $a2 k $a3 $lib: "
      env $e timeout -k 10 120 python3 $R/scripts/bench_batch_dim.py --rows $a1 --dim $a2 --k $a3 --steps 200 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step %.4f kernel %.4f' % (d['ms_per_step'], d['kernel_ms']))"
    done
  done
done
