"""Edge cases of the C ABI the reference's callers can hit (empty / tiny / degenerate inputs)."""
import uuid

import numpy as np
import pytest

from conftest import assert_topk_parity, ids_for

pytestmark = pytest.mark.gpu


def test_k_zero_and_empty_batches(hip, oracle):
    h = hip.HipIndex(8)
    assert h.search(np.ones(8, np.float32), 0) == []
    assert h.search_batch([], 5) == {}
    h.insert(uuid.uuid4(), np.ones(8, np.float32))
    assert h.search(np.ones(8, np.float32), 0) == []
    ids, s, d, c = h.search_batch_arrays(np.ones((3, 8), np.float32), 0)
    assert list(c) == [0, 0, 0]
    r = h.search(np.ones(8, np.float32), 2 ** 62)           # any k: clamps to the rows present
    assert len(r) == 1 and r[0].score > 0.999


def test_tiny_dimensions(hip, oracle):
    for d in (1, 2, 5):
        rng = np.random.default_rng(d)
        rows = rng.standard_normal((40, d)).astype(np.float32)
        ids = ids_for(40)
        h = hip.HipIndex(d)
        h.insert_batch(ids, rows)
        o = oracle.OracleIndex(d)
        o.insert_batch(ids, rows)
        q = rng.standard_normal(d).astype(np.float32)
        gi, gs, gd = h.search_arrays(q, 40)
        e = o.search(q, 40)
        lut = {ids[i].tobytes(): i for i in range(40)}
        assert_topk_parity([lut[x.tobytes()] for x in gi], gs, e["row"], e["score"], what=f"dim={d}")


def test_passes_on_empty_and_single_row_indexes(hip):
    h = hip.HipIndex(768)
    fr, to, w = h.autolink_pass_rows(None, 100, 0.75, 50)
    assert len(fr) == 0
    a, b, s = h.dedup_scan_rows(0.92)
    assert len(a) == 0
    h.insert(uuid.uuid4(), np.ones(768, np.float32))
    fr, to, w = h.autolink_pass_rows(None, 100, 0.75, 50)   # only itself: self is skipped
    assert len(fr) == 0
    a, b, s = h.dedup_scan_rows(0.92)
    assert len(a) == 0
    # two identical rows: one directed edge each way, one dedup pair
    h.insert(uuid.uuid4(), np.ones(768, np.float32))
    fr, to, w = h.autolink_pass_rows(None, 100, 0.75, 50)
    assert sorted(zip(fr.tolist(), to.tolist())) == [(0, 1), (1, 0)] and np.all(w > 0.9999)
    a, b, s = h.dedup_scan_rows(0.92)
    assert list(zip(a.tolist(), b.tolist())) == [(0, 1)]


def test_argument_errors_are_validation_errors(hip):
    h = hip.HipIndex(4)
    h.insert(uuid.uuid4(), np.ones(4, np.float32))
    with pytest.raises(hip.ValidationError):
        h.autolink_pass_rows(np.array([7], np.uint32), 100, 0.75, 50)      # scan row out of range
    with pytest.raises(hip.ValidationError):
        h.autolink_pass_rows(None, 1000, 0.75, 50)                          # topk above the list limit
    with pytest.raises(hip.ValidationError):
        h.insert(b"short", np.ones(4, np.float32))
    with pytest.raises(hip.ValidationError, match="Embedding dimension mismatch: expected 4, got 5"):
        h.insert_batch(ids_for(2), np.ones((2, 5), np.float32))
    assert len(h) == 1                                                      # a failed batch inserted nothing


def test_duplicate_ids_inside_one_batch_keep_the_last_vector(hip, oracle):
    ids = ids_for(3)
    batch_ids = np.stack([ids[0], ids[1], ids[0], ids[2], ids[1]])
    vecs = np.eye(5, dtype=np.float32)
    h = hip.HipIndex(5)
    h.insert_batch(batch_ids, vecs)
    assert len(h) == 3
    r = h.search(vecs[2], 1)      # id0 now holds e2
    assert r[0].node_id.bytes == ids[0].tobytes() and r[0].score > 0.999
    r = h.search(vecs[4], 1)      # id1 now holds e4
    assert r[0].node_id.bytes == ids[1].tobytes()
    assert h.row_id(0).bytes == ids[0].tobytes() and h.row_id(2).bytes == ids[2].tobytes()  # first-seen order


def test_growth_across_many_small_inserts(hip, oracle):
    d = 128
    rows = oracle.synth_rows(3000, d)
    ids = ids_for(3000)
    h = hip.HipIndex(d)               # starts at the minimum capacity and doubles
    o = oracle.OracleIndex(d)
    for lo in range(0, 3000, 333):
        h.insert_batch(ids[lo:lo + 333], rows[lo:lo + 333])
        o.insert_batch(ids[lo:lo + 333], rows[lo:lo + 333])
    q = oracle.synth_queries(3000, d, 1)[0]
    gi, gs, _ = h.search_arrays(q, 20)
    e = o.search(q, 20)
    lut = {ids[i].tobytes(): i for i in range(3000)}
    assert_topk_parity([lut[x.tobytes()] for x in gi], gs, e["row"], e["score"], what="after growth")
