#!/bin/bash
# same-box A/B of two builds over batched-search shapes: LIBS="a.so b.so" SHAPES="rows dim k dtype;..." bash scripts/ab_batch_shapes.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
IFS=';' read -ra SH <<< "${SHAPES:-6250000 1024 10 bf16;1000000 1024 10 f32;1250000 768 100 f32}"
for rep in 1 2; do for a in "${SH[@]}"; do set -- $a; for lib in ${LIBS:-libcortex_hip_base.so libcortex_hip.so}; do
  echo -n "rep $rep [$a] $lib: "; CORTEX_HIP_LIB=$R/cortex_amd/lib/$lib timeout -k 10 300 python3 $R/scripts/bench_batch_dim.py --rows $1 --dim $2 --k $3 --dtype $4 --steps 20 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('kernel', round(d['kernel_ms'],4), 'step', round(d['ms_per_step'],4), 'fixed', round(d['ms_per_step']-d['kernel_ms'],4))"
done; done; done
