"""The reference's own vector-layer tests (crates/cortex-core/src/vector/index.rs:484-728),
run against the HIP engine through the C ABI.  Same bodies, HipIndex for HnswIndex."""
import uuid

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def now_v7():
    return uuid.uuid4()


def test_index_insert_and_search(hip):  # :484-510
    index = hip.HipIndex.new(3)
    id1, id2, id3 = now_v7(), now_v7(), now_v7()
    index.insert(id1, [1.0, 0.0, 0.0])
    index.insert(id2, [0.9, 0.1, 0.0])
    index.insert(id3, [0.0, 1.0, 0.0])
    index.rebuild()
    results = index.search([1.0, 0.0, 0.0], 2, None)
    assert len(results) == 2
    assert results[0].node_id == id1


def test_threshold_search(hip):  # :513-535
    index = hip.HipIndex.new(3)
    id1, id2 = now_v7(), now_v7()
    index.insert(id1, [1.0, 0.0, 0.0])
    index.insert(id2, [0.0, 1.0, 0.0])
    index.rebuild()
    results = index.search_threshold([1.0, 0.0, 0.0], 0.95, None)
    assert len(results) == 1
    assert results[0].node_id == id1


def test_dimension_mismatch_rejected(hip):  # :579-583
    index = hip.HipIndex.new(3)
    with pytest.raises(hip.ValidationError, match="Embedding dimension mismatch: expected 3, got 2"):
        index.insert(now_v7(), [1.0, 2.0])


def test_empty_index_search(hip):  # :586-590
    index = hip.HipIndex.new(3)
    assert index.search([1.0, 0.0, 0.0], 5, None) == []
    assert index.is_empty()


def test_brute_force_fallback(hip):  # :593-606
    index = hip.HipIndex.new(3)
    id1, id2 = now_v7(), now_v7()
    index.insert(id1, [1.0, 0.0, 0.0])
    index.insert(id2, [0.0, 1.0, 0.0])
    results = index.search([1.0, 0.0, 0.0], 2, None)  # no rebuild
    assert len(results) == 2
    assert results[0].node_id == id1


def test_filter_by_kind(hip):  # :609-627
    index = hip.HipIndex.new(3)
    id1, id2 = now_v7(), now_v7()
    index.insert(id1, [1.0, 0.0, 0.0])
    index.set_metadata(id1, "fact", "test")
    index.insert(id2, [0.9, 0.1, 0.0])
    index.set_metadata(id2, "decision", "test")
    index.rebuild()
    f = hip.VectorFilter.new().with_kinds(["decision"])
    results = index.search([1.0, 0.0, 0.0], 5, f)
    assert len(results) == 1
    assert results[0].node_id == id2


def test_metadata_set_before_insert_binds(hip, tmp_path):  # vector/tests.rs:65-66: set_metadata, THEN insert
    index = hip.HipIndex.new(3)
    id1, id2, id3, id4 = now_v7(), now_v7(), now_v7(), now_v7()
    index.set_metadata(id1, "fact", "test")
    index.insert(id1, [1.0, 0.0, 0.0])
    index.set_metadata(id2, "decision", "test")
    index.insert(id2, [0.9, 0.1, 0.0])
    index.set_metadata(id3, "decision", "test")   # never gets a vector
    index.set_metadata(id4, "fact", "other")      # gets its vector after a save/load round trip
    f = hip.VectorFilter.new().with_kinds(["decision"])
    assert [r.node_id for r in index.search([1.0, 0.0, 0.0], 5, f)] == [id2]
    # a kind nobody was tagged with: rows with metadata fail, and the lookup does not grow the table (&self)
    assert index.search([1.0, 0.0, 0.0], 5, hip.VectorFilter.new().with_kinds(["never-seen"])) == []
    assert index.lookup("never-seen") == 0 and index.lookup("decision") == index.intern("decision") != 0
    # metadata of ids without a vector is part of the saved map (index.rs:438) and binds after load
    path = tmp_path / "meta.hnsw"
    index.save(path)
    loaded = hip.HipIndex.load(path)
    loaded.insert(id4, [0.7, 0.3, 0.0])
    fa = hip.VectorFilter.new().with_source_agent("other")
    assert [r.node_id for r in loaded.search([1.0, 0.0, 0.0], 5, fa)] == [id4]
    index.remove(id3)                              # metadata.remove without a vector (:318)
    index.insert(id3, [0.8, 0.2, 0.0])             # now a node without metadata: passes every kind filter (Q3)
    assert [r.node_id for r in index.search([1.0, 0.0, 0.0], 5, f)] == [id2, id3]


def test_filter_exclude(hip):  # :630-646
    index = hip.HipIndex.new(3)
    id1, id2 = now_v7(), now_v7()
    index.insert(id1, [1.0, 0.0, 0.0])
    index.insert(id2, [0.9, 0.1, 0.0])
    index.rebuild()
    f = hip.VectorFilter.new().excluding([id1])
    results = index.search([1.0, 0.0, 0.0], 5, f)
    assert len(results) == 1
    assert results[0].node_id == id2


def test_remove_doesnt_crash_search(hip):  # :649-664
    index = hip.HipIndex.new(3)
    id1, id2 = now_v7(), now_v7()
    index.insert(id1, [1.0, 0.0, 0.0])
    index.insert(id2, [0.0, 1.0, 0.0])
    index.rebuild()
    index.remove(id1)
    assert index.len() == 1
    results = index.search([1.0, 0.0, 0.0], 5, None)
    assert len(results) > 0
    assert all(r.node_id != id1 for r in results)  # exact engine: no stale hits (Q1/Q2)


def test_search_batch(hip):  # :667-684
    index = hip.HipIndex.new(3)
    id1, id2, id3 = now_v7(), now_v7(), now_v7()
    index.insert(id1, [1.0, 0.0, 0.0])
    index.insert(id2, [0.0, 1.0, 0.0])
    index.insert(id3, [0.0, 0.0, 1.0])
    index.rebuild()
    queries = [(id1, [1.0, 0.0, 0.0]), (id2, [0.0, 1.0, 0.0])]
    results = index.search_batch(queries, 1, None)
    assert len(results) == 2
    assert results[id1][0].node_id == id1
    assert results[id2][0].node_id == id2


def test_similarity_score_range(hip):  # :687-708
    index = hip.HipIndex.new(3)
    index.insert(now_v7(), [1.0, 0.0, 0.0])
    index.insert(now_v7(), [-1.0, 0.0, 0.0])
    index.rebuild()
    results = index.search([1.0, 0.0, 0.0], 2, None)
    for r in results:
        assert 0.0 <= r.score <= 1.0, f"Score {r.score} out of range"
    assert results[0].score > 0.99
    assert results[1].score == 0.0 and results[1].distance == 2.0


def test_threshold_returns_only_above(hip):  # :711-728
    index = hip.HipIndex.new(3)
    id_close, id_far = now_v7(), now_v7()
    index.insert(id_close, [1.0, 0.0, 0.0])
    index.insert(id_far, [0.0, 0.0, 1.0])
    index.rebuild()
    results = index.search_threshold([1.0, 0.0, 0.0], 0.5, None)
    assert all(r.score >= 0.5 for r in results)
    assert any(r.node_id == id_close for r in results)
