#!/bin/bash
# round 3, step 3: variants of the persistent kernel (CX_PAIR_P_VAR bit mask), interleaved A/B rounds in one call
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3; mkdir -p $O
cd $R
python -m pytest tests/test_hip_autolink.py -x -q -m gpu -k "persistent" > $O/step3_tests.log 2>&1; echo "tests rc=$?" >> $O/step3_tests.log; tail -3 $O/step3_tests.log
L=$O/step3.log; : > $L
run() { echo "== $*" >> $L; env "$@" timeout -k 10 200 python3 $R/scripts/bench_autolink.py --reps 10 2>&1 | grep -v amdgpu.ids >> $L; }
for round in 1 2; do
  for v in 0 1 2 4 5 6 7; do run CX_PAIR_PERSIST=1 CX_PAIR_P_DYN=1 CX_PAIR_P_VAR=$v CX_PAIR_P_CLOCK=1; done
done
run CX_PAIR_PERSIST=1 CX_PAIR_P_DYN=1 CX_PAIR_P_VAR=4 CX_PAIR_DIAG=1
python3 - <<'PY' $L
import sys, re, json
arm = None
for line in open(sys.argv[1]):
    if line.startswith("=="): arm = line.strip(); clk = []
    elif line.startswith("[pair_p] block 0"):
        m = re.search(r"in ([0-9.]+) ms = ([0-9.]+) GHz", line); clk.append((float(m.group(1)), float(m.group(2))))
    elif line.startswith("[pair_p diag]"): print(arm, line.strip())
    elif line.startswith("{"):
        j = json.loads(line); best = min(clk) if clk else (0, 0)
        print(f"{arm:80s} phase {j['phase_ms']['filter_gemm']:.3f} ms  kernel best {best[0]:.3f} ms @ {best[1]:.3f} GHz  -> {7.714e12 / (best[0] * 1e-3) / 2.5e15 if best[0] else 0:.3f}")
PY
