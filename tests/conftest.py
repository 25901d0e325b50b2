import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Score tolerance GPU vs oracle (north_star: "within a stated fp tolerance").
# The oracle sums d products sequentially in f32, the kernels sum lane-strided
# partials with a butterfly / MFMA: both are within d*2^-24*|a||b| of the exact
# value; for unit rows at d <= 1024 that is <= 6.1e-5 worst case, ~2e-6 typical.
SCORE_TOL = 5e-5


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available() -> bool:
    try:
        from cortex_amd import _lib
        return _lib.load().cx_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip: no silent fallback.
    pass


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def hip():
    import cortex_amd
    from cortex_amd import _lib
    L = _lib.load()
    assert L.cx_device_count() > 0, "gpu test on a machine without a HIP device"
    return cortex_amd


def ids_for(n: int, salt: int = 0) -> np.ndarray:
    """Deterministic 16-byte ids: UUIDv7-like layout is irrelevant to the engine; bytes are opaque."""
    rng = np.random.default_rng(1000 + salt)
    ids = rng.integers(0, 256, size=(n, 16), dtype=np.uint8)
    ids[:, 8:] = np.arange(n, dtype=np.uint64).astype(">u8").view(np.uint8).reshape(n, 8)  # unique
    return ids


def assert_topk_parity(got_rows, got_scores, exp_rows, exp_scores, tol=SCORE_TOL, what=""):
    """ids exact except where adjacent scores are closer than tol (near-ties are sets)."""
    got_rows = np.asarray(got_rows).astype(np.int64)
    exp_rows = np.asarray(exp_rows).astype(np.int64)
    got_scores = np.asarray(got_scores, dtype=np.float64)
    exp_scores = np.asarray(exp_scores, dtype=np.float64)
    assert len(got_rows) == len(exp_rows), f"{what}: result count {len(got_rows)} != {len(exp_rows)}"
    if len(exp_rows) == 0:
        return
    both_nan = np.isnan(got_scores) & np.isnan(exp_scores)
    diff = np.where(both_nan, 0.0, np.abs(got_scores - exp_scores))
    assert np.all(diff <= tol), f"{what}: score mismatch max {np.nanmax(diff)} > {tol}"
    pos = {int(r): j for j, r in enumerate(exp_rows)}
    for i, r in enumerate(got_rows):
        if int(r) in pos:
            j = pos[int(r)]
            if j != i:
                a, b = exp_scores[j], exp_scores[i]
                assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= tol, \
                    f"{what}: row {r} at rank {i}, oracle rank {j}, scores {a} vs {b} are not a near-tie"
        else:
            a, b = got_scores[i], exp_scores[-1]
            assert abs(a - b) <= tol, f"{what}: row {r} (score {a}) not in oracle list and not tied with its tail {b}"
