"""The reference's own known-answer tests for the hot path, re-expressed against the oracle.

Each test names the reference test it restates (crates/cortex-core/src/...).
These are what pins the CPU restatement: the reference asserts
ids, lengths and score ranges only — never a numeric score.
"""
import uuid

import numpy as np
import pytest


def new_id():
    return uuid.uuid4()


def test_index_insert_and_search(oracle):  # vector/index.rs:484-510
    ix = oracle.OracleIndex(3)
    id1, id2, id3 = new_id(), new_id(), new_id()
    ix.insert(id1, [1.0, 0.0, 0.0])
    ix.insert(id2, [0.9, 0.1, 0.0])
    ix.insert(id3, [0.0, 1.0, 0.0])
    ix.rebuild()
    r = ix.search([1.0, 0.0, 0.0], 2)
    assert len(r) == 2
    assert bytes(r[0]["node_id"]) == id1.bytes


def test_threshold_search(oracle):  # vector/index.rs:513-535
    ix = oracle.OracleIndex(3)
    id1, id2 = new_id(), new_id()
    ix.insert(id1, [1.0, 0.0, 0.0])
    ix.insert(id2, [0.0, 1.0, 0.0])
    ix.rebuild()
    r = ix.search_threshold([1.0, 0.0, 0.0], 0.95)
    assert len(r) == 1
    assert bytes(r[0]["node_id"]) == id1.bytes


def test_dimension_mismatch_rejected(oracle):  # vector/index.rs:579-583
    ix = oracle.OracleIndex(3)
    with pytest.raises(oracle.OracleError, match="Embedding dimension mismatch: expected 3, got 2"):
        ix.insert(new_id(), [1.0, 2.0])


def test_empty_index_search(oracle):  # vector/index.rs:586-590
    ix = oracle.OracleIndex(3)
    assert len(ix.search([1.0, 0.0, 0.0], 5)) == 0


def test_brute_force_fallback(oracle):  # vector/index.rs:593-606
    ix = oracle.OracleIndex(3)
    id1, id2 = new_id(), new_id()
    ix.insert(id1, [1.0, 0.0, 0.0])
    ix.insert(id2, [0.0, 1.0, 0.0])
    r = ix.search([1.0, 0.0, 0.0], 2)
    assert len(r) == 2
    assert bytes(r[0]["node_id"]) == id1.bytes


def test_filter_by_kind(oracle):  # vector/index.rs:609-627
    ix = oracle.OracleIndex(3)
    id1, id2 = new_id(), new_id()
    ix.insert(id1, [1.0, 0.0, 0.0])
    ix.set_metadata(id1, "fact", "test")
    ix.insert(id2, [0.9, 0.1, 0.0])
    ix.set_metadata(id2, "decision", "test")
    ix.rebuild()
    r = ix.search([1.0, 0.0, 0.0], 5, oracle.Filter(kinds=["decision"]))
    assert len(r) == 1
    assert bytes(r[0]["node_id"]) == id2.bytes


def test_metadata_set_before_insert_binds(oracle):  # vector/tests.rs:65-66: set_metadata, THEN insert
    """`metadata` is a map of its own (index.rs:189, :219-222): an entry made before the vector exists is found by
    matches_filter (:234) once the vector is there; remove drops it (:318) even if no vector ever came."""
    ix = oracle.OracleIndex(3)
    id1, id2, id3 = new_id(), new_id(), new_id()
    ix.set_metadata(id1, "fact", "test")
    ix.insert(id1, [1.0, 0.0, 0.0])
    ix.set_metadata(id2, "decision", "test")
    ix.insert(id2, [0.9, 0.1, 0.0])
    ix.set_metadata(id3, "decision", "test")     # never gets a vector
    r = ix.search([1.0, 0.0, 0.0], 5, oracle.Filter(kinds=["decision"]))
    assert [bytes(x["node_id"]) for x in r] == [id2.bytes]
    ix.remove(id3)                                # metadata.remove without a vector
    ix.insert(id3, [0.8, 0.2, 0.0])               # now a node without metadata: passes every kind filter (Q3)
    r = ix.search([1.0, 0.0, 0.0], 5, oracle.Filter(kinds=["decision"]))
    assert [bytes(x["node_id"]) for x in r] == [id2.bytes, id3.bytes]


def test_filter_exclude(oracle):  # vector/index.rs:630-646
    ix = oracle.OracleIndex(3)
    id1, id2 = new_id(), new_id()
    ix.insert(id1, [1.0, 0.0, 0.0])
    ix.insert(id2, [0.9, 0.1, 0.0])
    ix.rebuild()
    r = ix.search([1.0, 0.0, 0.0], 5, oracle.Filter(exclude=[id1.bytes]))
    assert len(r) == 1
    assert bytes(r[0]["node_id"]) == id2.bytes


def test_remove_doesnt_crash_search(oracle):  # vector/index.rs:649-664
    ix = oracle.OracleIndex(3)
    id1, id2 = new_id(), new_id()
    ix.insert(id1, [1.0, 0.0, 0.0])
    ix.insert(id2, [0.0, 1.0, 0.0])
    ix.rebuild()
    ix.remove(id1)
    assert len(ix) == 1
    assert len(ix.search([1.0, 0.0, 0.0], 5)) > 0


def test_search_batch(oracle):  # vector/index.rs:667-684
    ix = oracle.OracleIndex(3)
    ids = [new_id() for _ in range(3)]
    for i, v in zip(ids, ([1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0])):
        ix.insert(i, v)
    ix.rebuild()
    res = ix.search_batch(np.array([[1.0, 0, 0], [0, 1.0, 0]], np.float32), 1, n_threads=2)
    assert len(res) == 2
    assert bytes(res[0][0]["node_id"]) == ids[0].bytes
    assert bytes(res[1][0]["node_id"]) == ids[1].bytes


def test_similarity_score_range(oracle):  # vector/index.rs:687-708
    ix = oracle.OracleIndex(3)
    ix.insert(new_id(), [1.0, 0.0, 0.0])
    ix.insert(new_id(), [-1.0, 0.0, 0.0])
    ix.rebuild()
    r = ix.search([1.0, 0.0, 0.0], 2)
    assert all(0.0 <= s <= 1.0 for s in r["score"])
    assert r[0]["score"] > 0.99
    assert r[1]["score"] == 0.0 and r[1]["distance"] == 2.0  # clamp on score only (Q7)


def test_threshold_returns_only_above(oracle):  # vector/index.rs:711-728
    ix = oracle.OracleIndex(3)
    close, far = new_id(), new_id()
    ix.insert(close, [1.0, 0.0, 0.0])
    ix.insert(far, [0.0, 0.0, 1.0])
    ix.rebuild()
    r = ix.search_threshold([1.0, 0.0, 0.0], 0.5)
    assert all(s >= 0.5 for s in r["score"])
    assert any(bytes(x["node_id"]) == close.bytes for x in r)


def test_default_config(oracle):  # vector/config.rs:93-103
    c = oracle.SimilarityConfig.default()
    assert c.auto_link_threshold == np.float32(0.75)
    assert c.dedup_threshold == np.float32(0.92)
    assert c.contradiction_threshold == np.float32(0.80)
    assert c.auto_link_k == 20
    c.validate()


def test_invalid_config(oracle):  # vector/config.rs:117-124
    c = oracle.SimilarityConfig(auto_link_threshold=0.95, dedup_threshold=0.90)
    with pytest.raises(oracle.OracleError, match="auto_link_threshold must be less than dedup_threshold"):
        c.validate()


def test_clamping(oracle):  # vector/config.rs:126-134
    assert oracle.clamp_threshold(1.5) == 1.0
    assert oracle.clamp_threshold(-0.5) == 0.0


def test_similarity_link_rule_threshold(oracle):  # linker/rules.rs:403-421: 0.8 passes, 0.5 fails at 0.75, `>=`
    ix = oracle.OracleIndex(2)
    a = np.array([1.0, 0.0], np.float32)
    for cos in (0.8, 0.5):
        ix.insert(new_id(), a if cos == 0.8 else [cos, np.sqrt(1 - cos * cos)])
    ix.insert(new_id(), [0.8, 0.6])
    edges = ix.autolink_pass([0], 100, 0.75, 50)
    assert [int(e["to_row"]) for e in edges] == [2]
    assert abs(edges[0]["weight"] - 0.8) < 1e-6
