#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_hip_bf16_store.py -x -q -m gpu -k "batch" > $O/bs_tests.log 2>&1; echo "tests rc=$?" >> $O/bs_tests.log; tail -3 $O/bs_tests.log
for a in "6250000 1024 10 bf16" "6250000 1024 10 f32" "1000000 1024 10 f32" "1250000 384 10 f32" "1250000 768 10 f32" "1250000 768 100 f32" "1250000 384 100 f32" "1000000 1024 100 f32"; do set -- $a; timeout -k 10 200 python3 scripts/bench_batch_dim.py --rows $1 --dim $2 --k $3 --dtype $4 --steps 20 2>/dev/null; done | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    r = json.loads(l); by = r['rows'] * r['dim'] * (2 if r['dtype'] == 'bf16' else 4)
    print(r['rows'], r['dim'], r['dtype'], 'k', r['k'], 'step ms', round(r['ms_per_step'], 3), 'kernel ms', round(r['kernel_ms'], 3), 'frac kernel', round(r['frac_of_8TBs'], 3), 'frac step', round(by / (r['ms_per_step'] * 1e-3) / 8e12, 3))
"
