#!/bin/bash
# round 3, step 1: persistent filter kernel — parity, then timing arms
set -o pipefail
mkdir -p gpurun_out/r3
python -m pytest tests/test_hip_autolink.py -x -q -m gpu -k "persistent or autolink_pass_matches or rescan" > gpurun_out/r3/step1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3/step1_tests.log
tail -5 gpurun_out/r3/step1_tests.log
for arm in "CX_PAIR_PERSIST=0" "CX_PAIR_PERSIST=1" "CX_PAIR_PERSIST=1 CX_PAIR_P_DYN=1" "CX_PAIR_PERSIST=1 CX_PAIR_P_WAVES=4" "CX_PAIR_PERSIST=0" "CX_PAIR_PERSIST=1"; do
  echo "== $arm" >> gpurun_out/r3/step1_bench.log
  env $arm timeout -k 10 300 python scripts/bench_autolink.py --reps 8 >> gpurun_out/r3/step1_bench.log 2>&1
done
cat gpurun_out/r3/step1_bench.log
