#!/bin/bash
# A/B of two builds of the library on one box: CORTEX_HIP_LIB picks the build, CX_BATCH_ARM the measurement arm
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do for d in ${DIMS:-384 768}; do for lib in ${LIBS:-libcortex_hip.so libcortex_hip_b.so}; do for arm in ${ARMS:-0 1}; do
  echo -n "rep $rep dim $d $lib arm $arm: "; CORTEX_HIP_LIB=$R/cortex_amd/lib/$lib CX_BATCH_ARM=$arm timeout -k 10 200 python3 $R/scripts/bench_batch_dim.py --rows ${ROWS:-1250000} --dim $d --k ${K:-10} --steps 20 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['kernel_ms'],4), round(d['ms_per_step'],4))"
done; done; done; done
