#!/bin/bash
# round 3, step 2: where the persistent filter kernel's time goes — phase stamps, clock, PMC, tile order
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
L=$O/step2.log; : > $L
run() { echo "== $*" >> $L; env "$@" timeout -k 10 200 python3 $R/scripts/bench_autolink.py --reps 6 >> $L 2>&1; }
run CX_PAIR_PERSIST=1 CX_PAIR_P_DYN=1 CX_PAIR_DIAG=1
run CX_PAIR_PERSIST=0 CX_PAIR_DIAG=1
run CX_PAIR_PERSIST=1 CX_PAIR_P_DYN=1 CX_PAIR_P_CLOCK=1
run CX_PAIR_PERSIST=1 CX_PAIR_P_DYN=0 CX_PAIR_P_CLOCK=1
for gs in 2 4 6 8 12 16; do run CX_PAIR_PERSIST=1 CX_PAIR_P_DYN=1 CX_PAIR_GS=$gs; done
run CX_PAIR_PERSIST=0
run CX_PAIR_PERSIST=1 CX_PAIR_P_DYN=1
# PMC (counters only, program directly after --)
for arm in 0 1; do
  export CX_PAIR_PERSIST=$arm CX_PAIR_P_DYN=1
  timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq_$arm -- python3 $R/scripts/bench_autolink.py --reps 2 > $O/pmc_sq_$arm.json 2> $O/pmc_sq_$arm.err || tail -3 $O/pmc_sq_$arm.err
  python3 $R/scripts/pmc_summary.py $O/pmc_sq_$arm pair_filter > $O/pmc_sq_${arm}_summary.json
  for gs in 4 8; do
    export CX_PAIR_GS=$gs
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_${arm}_$gs -- python3 $R/scripts/bench_autolink.py --reps 2 > /dev/null 2> $O/pmc_fetch.err || tail -3 $O/pmc_fetch.err
    python3 $R/scripts/pmc_summary.py $O/pmc_fetch_${arm}_$gs pair_filter > $O/pmc_fetch_${arm}_gs${gs}_summary.json
    rm -rf $O/pmc_fetch_${arm}_$gs
  done
  unset CX_PAIR_GS
  rm -rf $O/pmc_sq_$arm
done
grep -v amdgpu.ids $L
cat $O/pmc_sq_*_summary.json $O/pmc_fetch_*_summary.json
