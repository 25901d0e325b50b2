"""cx_bulk_load_nodes: the reference's start-up loop over stored nodes
(cortex-server/src/serve.rs:105-123, cortex-core/src/api.rs:56-70, list_nodes storage/redb_storage.rs:705-734)
restated in Python over the same records, against the one-call loader."""
import uuid

import numpy as np
import pytest

import bincode_ref as B
from conftest import assert_topk_parity

pytestmark = pytest.mark.gpu


def _ts(sec, ns=0):
    import datetime
    t = datetime.datetime(2024, 1, 1, tzinfo=datetime.timezone.utc) + datetime.timedelta(seconds=int(sec))
    frac = "" if ns == 0 else ".%09d" % ns
    return t.strftime("%Y-%m-%dT%H:%M:%S") + frac + "Z"


def _records(oracle, n, d, seed):
    rng = np.random.default_rng(seed)
    rows = oracle.synth_rows(n, d)
    recs, nodes = [], []
    for i in range(n):
        u = rng.random()
        emb = rows[i]
        if u < 0.10:
            emb = None
        elif u < 0.15:
            emb = rows[i][: d - 1]            # wrong length: insert fails
        sec, ns = int(rng.integers(0, 40)), int(rng.choice([0, 0, 5, 999999999]))  # many equal created_at: stable order matters
        node = dict(id16=uuid.UUID(int=int(rng.integers(1, 1 << 62)) << 32 | i).bytes, kind="fact" if i % 3 else "decision",
                    title="t%d" % i, body="b" * int(rng.integers(0, 9)), tags=[], embedding=emb, agent="agent-%d" % (i % 4),
                    session=None, channel=None, importance=0.5, access_count=i, last_accessed_at=_ts(0),
                    created_at=_ts(sec, ns), updated_at=_ts(sec), deleted=bool(rng.random() < 0.1))
        rec = B.encode_node(**node)
        if rng.random() < 0.05:
            rec = rec[: len(rec) // 2]        # corrupt record: list_nodes skips it
            node = None
        nodes.append((node, (sec, ns)))
        recs.append(rec)
    return recs, nodes


def _reference_loop(nodes, d, include_deleted=False):
    """list_nodes(NodeFilter::new()) then the insert loop; returns [(id, embedding)] in insertion order + counters."""
    st = dict(records=len(nodes), undecodable=0, deleted=0, no_embedding=0, dim_mismatch=0, indexed=0)
    listed = []
    for node, key in nodes:
        if node is None:
            st["undecodable"] += 1
        elif node["deleted"] and not include_deleted:
            st["deleted"] += 1
        else:
            listed.append((node, key))
    listed.sort(key=lambda t: t[1], reverse=True)   # Python's sort is stable, like slice::sort_by; reverse keeps ties in order
    out = []
    for node, _ in listed:
        if node["embedding"] is None:
            st["no_embedding"] += 1
        elif len(node["embedding"]) != d:
            st["dim_mismatch"] += 1
        else:
            out.append((node["id16"], node["embedding"], node))
            st["indexed"] += 1
    return out, st


def test_sort_reverse_is_stable_for_ties():
    # the restated loop relies on it: equal keys keep their table order under reverse=True
    a = [((1, 0), "x"), ((1, 0), "y"), ((2, 0), "z")]
    assert [v for _, v in sorted(a, key=lambda t: t[0], reverse=True)] == ["z", "x", "y"]


@pytest.mark.parametrize("include_deleted", [False, True])
def test_bulk_load_matches_the_reference_loop(hip, oracle, include_deleted):
    n, d = 700, 384
    recs, nodes = _records(oracle, n, d, seed=11)
    want, st_want = _reference_loop(nodes, d, include_deleted)
    h = hip.HipIndex(d)
    st = h.bulk_load_nodes(recs, include_deleted=include_deleted)
    assert st == st_want and st["undecodable"] > 0 and st["dim_mismatch"] > 0 and st["no_embedding"] > 0
    assert h.len() == len(want)
    # rows in list_nodes order (newest first, ties in table order)
    assert [h.row_id(r).bytes for r in range(len(want))] == [w[0] for w in want]
    # and the loaded index answers like one filled by the insert loop
    g = hip.HipIndex(d)
    for i, e, _ in want:
        g.insert(i, e)
    q = oracle.synth_queries(n, d, 8)
    for qi in range(8):
        a, b = h.search(q[qi], 10, None), g.search(q[qi], 10, None)
        assert [r.node_id for r in a] == [r.node_id for r in b]
        assert [r.score for r in a] == [r.score for r in b]


def test_bulk_load_parity_with_oracle_search(hip, oracle):
    n, d = 500, 384
    recs, nodes = _records(oracle, n, d, seed=12)
    want, _ = _reference_loop(nodes, d)
    h = hip.HipIndex(d)
    h.bulk_load_nodes(recs)
    o = oracle.OracleIndex(d)            # the oracle index filled by the reference's loop
    for i, e, _ in want:
        o.insert(i, e)
    lut = {w[0]: r for r, w in enumerate(want)}
    q = oracle.synth_queries(n, d, 4)
    for qi in range(4):
        res = h.search(q[qi], 10, None)
        exp = o.search(q[qi], 10)
        got_rows = np.array([lut[r.node_id.bytes] for r in res])
        assert_topk_parity(got_rows, np.array([r.score for r in res]), exp["row"], exp["score"])


def test_strict_is_cortex_open(hip, oracle):
    d = 384
    recs, nodes = _records(oracle, 200, d, seed=13)
    h = hip.HipIndex(d)
    with pytest.raises(hip.ValidationError, match=r"Embedding dimension mismatch: expected 384, got 383"):
        h.bulk_load_nodes(recs, strict=True)


def test_set_metadata_flag_binds_filters(hip, oracle):
    d = 384
    recs, nodes = _records(oracle, 300, d, seed=14)
    want, _ = _reference_loop(nodes, d)
    h = hip.HipIndex(d)
    h.bulk_load_nodes(recs, set_metadata=True)
    q = want[0][1]
    res = h.search(q, 50, hip.VectorFilter(kinds=["decision"], source_agent="agent-2"))
    by_id = {w[0]: w[2] for w in want}
    assert res and all(by_id[r.node_id.bytes]["kind"] == "decision" and by_id[r.node_id.bytes]["agent"] == "agent-2" for r in res)
    exp = sum(1 for w in want if w[2]["kind"] == "decision" and w[2]["agent"] == "agent-2")
    assert len(res) == min(50, exp)
    # without the flag the reference's behaviour: no metadata, every filter passes
    g = hip.HipIndex(d)
    g.bulk_load_nodes(recs)
    assert len(g.search(q, 50, hip.VectorFilter(kinds=["decision"], source_agent="agent-2"))) == 50


def test_keep_order_and_empty_input(hip, oracle):
    d = 384
    recs, nodes = _records(oracle, 120, d, seed=15)
    h = hip.HipIndex(d)
    assert h.bulk_load_nodes([]) == dict(records=0, undecodable=0, deleted=0, no_embedding=0, dim_mismatch=0, indexed=0)
    st = h.bulk_load_nodes(recs, keep_order=True)
    kept = [nd["id16"] for nd, _ in nodes if nd and not nd["deleted"] and nd["embedding"] is not None and len(nd["embedding"]) == d]
    assert st["indexed"] == len(kept) and [h.row_id(r).bytes for r in range(len(kept))] == kept
