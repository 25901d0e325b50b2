"""The on-device synthetic generator (bench support) is bit-identical to its CPU twin,
so full-size corpora generated in HBM are the same data the oracle sees at small sizes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_total,d,lo,n,flags", [
    (4000, 96, 0, 4000, 1), (4000, 96, 990, 20, 1), (3000, 768, 0, 3000, 1), (2500, 384, 0, 2500, 3),
    (100, 7, 0, 100, 0),
])
def test_device_generator_matches_cpu_twin(hip, oracle, n_total, d, lo, n, flags):
    import torch
    from cortex_amd import _lib
    L = _lib.load()
    out = torch.empty((n, d), dtype=torch.float32, device="cuda:0")
    rc = L.cx_synth_fill_dev(0, out.data_ptr(), oracle.SEED_CORPUS, oracle.SEED_CORPUS, oracle.SEED_DUP,
                             max(1, n_total // 50), lo, n, d, flags)
    assert rc == 0, L.cx_last_error()
    want = oracle.synth_rows(n_total, d, lo, n, flags=flags)
    got = out.cpu().numpy()
    assert np.array_equal(got, want), f"max diff {np.abs(got - want).max()}"
