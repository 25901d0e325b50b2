"""VectorIndex::save / load (vector/index.rs:437-473) through the C ABI, in the reference's file format."""
import uuid

import numpy as np
import pytest

import bincode_ref as B
from conftest import assert_topk_parity, ids_for

pytestmark = pytest.mark.gpu


def test_index_persistence(hip, tmp_path):  # vector/index.rs:538-566
    index_path = tmp_path / "test.hnsw"
    index = hip.HipIndex.new(3)
    id1 = uuid.uuid4()
    index.insert(id1, [1.0, 0.0, 0.0])
    index.rebuild()
    index.save(index_path)
    loaded_index = hip.HipIndex.load(index_path)
    assert loaded_index.len() == 1
    results = loaded_index.search([1.0, 0.0, 0.0], 1, None)
    assert len(results) == 1
    assert results[0].node_id == id1


def test_saved_file_is_the_reference_layout(hip, oracle, tmp_path):
    n, d = 300, 384
    rows = oracle.synth_rows(n, d)
    ids = ids_for(n)
    h = hip.HipIndex(d)
    h.insert_batch(ids, rows)
    for r in range(0, n, 7):
        h.set_metadata(ids[r].tobytes(), "fact" if r % 2 else "decision", f"agent-{r % 3}")
    h.remove(ids[5].tobytes())                       # removed rows are not written
    p = tmp_path / "ix.bin"
    h.save(p)
    vecs, meta, dim = B.decode_index_file(p.read_bytes())
    assert dim == d and len(vecs) == n - 1
    keep = [r for r in range(n) if r != 5]
    for (i, v), r in zip(vecs, keep):                # row order
        assert i == ids[r].tobytes() and np.array_equal(v, rows[r])
    assert meta == {ids[r].tobytes(): ("fact" if r % 2 else "decision", f"agent-{r % 3}") for r in range(0, n, 7) if r != 5}


def test_load_a_file_written_in_the_reference_layout(hip, oracle, tmp_path):
    n, d = 500, 768
    rows = oracle.synth_rows(n, d)
    ids = ids_for(n)
    rng = np.random.default_rng(1)
    order = rng.permutation(n)                       # a HashMap's arbitrary order
    vecs = [(ids[r].tobytes(), rows[r]) for r in order]
    meta = {ids[r].tobytes(): ("event", "kai") for r in range(0, n, 5)}
    p = tmp_path / "ref.bin"
    p.write_bytes(B.encode_index_file(vecs, meta, d))
    h = hip.HipIndex.load(p)
    assert len(h) == n and h.dimension == d
    o = oracle.OracleIndex(d)
    o.insert_batch(ids[order], rows[order])          # same insertion order as the file
    for r in range(0, n, 5):
        o.set_metadata(ids[r].tobytes(), "event", "kai")
    q = oracle.synth_queries(n, d, 3)
    lut = {ids[order[i]].tobytes(): i for i in range(n)}
    for qq in q:
        for flt_h, flt_o in ((None, None), (hip.VectorFilter(kinds=["event"]), oracle.Filter(kinds=["event"]))):
            gi, gs, gd = h.search_arrays(qq, 10, flt_h)
            e = o.search(qq, 10, flt_o)
            assert_topk_parity([lut[x.tobytes()] for x in gi], gs, e["row"], e["score"], what="loaded")
    # round trip: save what was loaded, decode, same content
    p2 = tmp_path / "again.bin"
    h.save(p2)
    v2, m2, d2 = B.decode_index_file(p2.read_bytes())
    assert d2 == d and m2 == meta and [x[0] for x in v2] == [x[0] for x in vecs]


def test_empty_index_and_errors(hip, tmp_path):
    h = hip.HipIndex(16)
    p = tmp_path / "empty.bin"
    h.save(p)
    assert B.decode_index_file(p.read_bytes()) == ([], {}, 16)
    h2 = hip.HipIndex.load(p)
    assert len(h2) == 0 and h2.dimension == 16
    with pytest.raises(hip.ValidationError, match="Failed to read index file"):
        hip.HipIndex.load(tmp_path / "missing.bin")
    bad = tmp_path / "bad.bin"
    bad.write_bytes(p.read_bytes()[:-3])
    with pytest.raises(hip.ValidationError, match="Failed to deserialize index"):
        hip.HipIndex.load(bad)
    with pytest.raises(hip.CortexError, match="Failed to write index file"):
        h.save(tmp_path / "no" / "such" / "dir" / "x.bin")
