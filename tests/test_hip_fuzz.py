"""Short runs of the randomised differential scripts (scripts/fuzz_*.py) as part of the GPU suite: random sizes, dims,
k, query counts, filters, tombstones, mutation histories and auto-link passes against the CPU oracle.  The scripts run
longer by hand (`--seconds`); the long runs of this round are recorded in profiles/r01/tuning.md."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,args,env", [
    ("fuzz_parity.py", ["--seconds", "6", "--seed", "101"], {}),
    ("fuzz_parity.py", ["--big", "--seconds", "7", "--seed", "102"], {}),
    ("fuzz_parity.py", ["--seconds", "5", "--seed", "105"], {"CX_BATCH_WIDE_MIN_K": "8"}),     # wide lists from k = 8
    # batchg.hip's bound + candidates path at fuzz sizes, 384/768-d included, with the fuzzer's filters and tombstones
    ("fuzz_parity.py", ["--seconds", "7", "--seed", "107"], {"CX_BATCHG_FILTER_MIN": "300", "CX_BATCH2": "0"}),
    ("fuzz_parity.py", ["--seconds", "5", "--seed", "108"], {"CX_BATCHG_FILTER_MIN": "300", "CX_BATCHG_SAMPLE_STEP": "7", "CX_BATCHG_CAND_CAP": "1"}),  # short lists: overflow -> dense fallback
    ("fuzz_stateful.py", ["--seconds", "5", "--seed", "109"], {"CX_BATCHG_FILTER_MIN": "300", "CX_BATCH2": "0"}),
    # bf16 row stores: the same fuzzers, the oracle fed the rounded rows
    ("fuzz_parity.py", ["--seconds", "6", "--seed", "110", "--dtype", "bf16"], {}),
    ("fuzz_parity.py", ["--seconds", "5", "--seed", "111", "--dtype", "bf16"], {"CX_BATCHG_FILTER_MIN": "300"}),
    ("fuzz_stateful.py", ["--seconds", "5", "--seed", "112", "--dtype", "bf16"], {}),
    ("fuzz_autolink.py", ["--seconds", "6", "--seed", "113", "--dtype", "bf16"], {}),
    ("fuzz_autolink.py", ["--seconds", "7", "--seed", "103"], {}),
    ("fuzz_autolink.py", ["--seconds", "6", "--seed", "106"], {"CX_PAIR_CAND_CAP": "24"}),     # most rows on the exact path
    ("fuzz_stateful.py", ["--seconds", "6", "--seed", "104"], {}),
])
def test_randomised_differential_run(hip, script, args, env):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script)] + args, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, **env))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "agree with the oracle" in r.stdout
