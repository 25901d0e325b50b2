"""The oracle on non-finite and extreme-magnitude vectors (CPU only): the C restatement (oracle/cortex_oracle.c) against the
independent numpy restatement (oracle/np_twin.py), bit for bit — the two were written separately from vector/index.rs:169-179,
:254-256, :259-294, and the GPU's handling of such vectors (tests/test_hip_irregular.py) is pinned to the former.

What the reference's arithmetic does with them (index.rs:172-177): a vector whose squares overflow -> norm = inf -> similarity
dot / inf = 0 -> score 0 (NaN if the dot overflowed too); squares that underflow -> norm 0 -> dot / 0 = +-inf -> distance -+inf ->
score 1.0 or 0.0; an Inf / NaN element or the zero vector -> NaN, sorted last (conftest.py's declared order)."""
import numpy as np

from conftest import ids_for


def _fixture(oracle, n=400, d=96, seed=4):
    rows = oracle.synth_rows(n, d).copy()
    qs = oracle.synth_queries(n, d, 12).copy()
    with np.errstate(over="ignore", under="ignore", invalid="ignore"):
        rows[3] *= np.float32(1e20)
        rows[4] *= np.float32(-1e20)
        rows[5] *= np.float32(1e-25)
        rows[6, 2] = np.inf
        rows[7, d - 1] = -np.inf
        rows[8, 0] = np.nan
        rows[9] = 0.0
        rows[10] *= np.float32(1e-16)     # |x|^2 = 1e-32: still exact in f32
        rows[11] = (rows[11] * np.float32(1e-25)) * np.float32(1e-16)   # elements ~1e-42: f32 DENORMALS (the fuzz's seed 97, case 177)
        qs[6] = rows[5]
        qs[7, 3] = np.nan
        qs[8] = 0.0
        qs[9] *= np.float32(1e20)
        qs[10] *= np.float32(1e-25)
        qs[11, 0] = np.inf
    return rows, qs


def test_c_oracle_equals_numpy_twin_on_irregular_vectors(oracle):
    from oracle import np_twin as T
    rows, qs = _fixture(oracle)
    o = oracle.OracleIndex(rows.shape[1])
    o.insert_batch(ids_for(len(rows)), rows)
    for qi, q in enumerate(qs):
        for k in (1, 10, len(rows)):
            e = o.search(q, k)
            idx, sc, di = T.brute_force(q, rows, k)
            assert list(map(int, e["row"])) == list(map(int, idx)), f"q{qi} k={k}: rows differ"
            assert np.array_equal(e["score"], sc, equal_nan=True), f"q{qi} k={k}: scores differ"
            assert np.array_equal(e["distance"], di, equal_nan=True), f"q{qi} k={k}: distances differ"


def test_what_the_reference_arithmetic_gives_for_each_kind(oracle):
    rows, qs = _fixture(oracle)
    o = oracle.OracleIndex(rows.shape[1])
    o.insert_batch(ids_for(len(rows)), rows)
    e = o.search(qs[0], len(rows))               # a regular query against everything
    by_row = {int(r): (float(s), float(d)) for r, s, d in zip(e["row"], e["score"], e["distance"])}
    assert by_row[3] == (0.0, 1.0) and by_row[4] == (0.0, 1.0), "squares overflow: dot / inf = 0"
    s5, d5 = by_row[5]
    assert (s5, d5) in ((1.0, -np.inf), (0.0, np.inf)), "squares underflow: dot / 0 = +-inf, clamped"
    s11, d11 = by_row[11]
    assert (s11, d11) in ((1.0, -np.inf), (0.0, np.inf)), "a row of denormals: its dot is a denormal, not 0 — +-inf like any underflowed norm"
    for r in (6, 7, 8, 9):
        assert np.isnan(by_row[r][0]) and np.isnan(by_row[r][1]), f"row {r}: NaN"
    assert list(map(int, e["row"][-4:])) == [6, 7, 8, 9], "NaN scores come last, in insertion order"
    assert by_row[10][0] > 0.0 or by_row[10][1] >= 1.0      # a small but representable norm: an ordinary cosine
    # the queries: NaN element / zero vector -> every score NaN, rows in insertion order; overflowing squares -> every finite
    # row ties at 0; underflowing squares -> half the store ties at 1.0
    for qi in (7, 8, 11):
        e = o.search(qs[qi], 5)
        assert np.all(np.isnan(e["score"])) and list(map(int, e["row"])) == [0, 1, 2, 3, 4]
    e = o.search(qs[9], 3)
    assert np.all(e["score"] == 0.0) and list(map(int, e["row"])) == [0, 1, 2]
    e = o.search(qs[10], 50)
    assert np.all(e["score"][:20] == 1.0) and np.all(np.diff(e["row"][:20].astype(np.int64)) > 0)
