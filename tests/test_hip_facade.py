"""`Cortex::{open, store, search}` (api.rs:50-131) over the HIP index, with the reference's own fakes for what is out of
scope: an in-memory storage and a deterministic embedder (the pattern of MockEmbedder / NoopIndex, briefing/engine.rs:
785-855, ingest.rs:224-293).  The facade's answers must be those of the same call sequence over the oracle's index."""
import hashlib
import uuid

import numpy as np
import pytest

from conftest import SCORE_TOL

pytestmark = pytest.mark.gpu

DIM = 384


class MemStorage:
    def __init__(self):
        self.nodes = {}
        self.order = []

    def put_node(self, node):
        if node.id not in self.nodes:
            self.order.append(node.id)
        self.nodes[node.id] = node

    def get_node(self, node_id):
        n = self.nodes.get(node_id)
        return None if n is None or n.deleted else n     # the facade's `if let Some(node)`; tombstoned nodes read as gone here

    def list_nodes(self):
        return [self.nodes[i] for i in reversed(self.order) if not self.nodes[i].deleted]   # newest first (redb_storage.rs:727-728)


class HashEmbedder:
    """text -> unit vector, deterministic; texts sharing words land near each other"""
    def dimension(self):
        return DIM

    def embed(self, text):
        v = np.zeros(DIM, np.float64)
        for w in text.lower().replace("\n", " ").replace(",", " ").replace(":", " ").split():
            seed = int.from_bytes(hashlib.sha256(w.encode()).digest()[:8], "little")
            v += np.random.default_rng(seed).standard_normal(DIM)
        v /= np.linalg.norm(v) or 1.0
        return v.astype(np.float32)


@pytest.mark.parametrize("devices", [None, [0, 0, 0]])
def test_facade_open_store_search(hip, oracle, devices):
    from cortex_amd.api import Cortex, Node, embedding_input
    emb = HashEmbedder()
    st = MemStorage()
    topics = ["jwt auth api token", "postgres index vacuum", "rust borrow checker lifetime", "pasta recipe garlic", "gpu kernel wavefront lds"]
    pre = []
    for i in range(60):
        n = Node(uuid.uuid4(), "fact", title=f"{topics[i % 5]} note {i}", body=f"details about {topics[i % 5]} number {i}", tags=[topics[i % 5].split()[0]])
        if i % 7:
            n.embedding = emb.embed(embedding_input(n))       # some stored nodes have no embedding yet (ensure_embedding's job)
        st.put_node(n)
        pre.append(n)
    cx = Cortex.open(st, emb, devices)
    o = oracle.OracleIndex(DIM)
    for n in st.list_nodes():                                 # the same insertion order as open()
        if n.embedding is not None:
            o.insert(n.id.bytes, n.embedding)
    seen = []
    cx.add_hook(lambda node, action: seen.append((node.id, action)))
    for i in range(25):                                       # store(): embedded on the way in
        n = Node(uuid.uuid4(), "decision", title=f"{topics[i % 5]} choice {i}", body=f"we picked {topics[(i + 1) % 5]}", tags=["t"])
        assert cx.store(n) == n.id
        o.insert(n.id.bytes, emb.embed(embedding_input(n)))
    assert len(seen) == 25 and seen[0][1] == "Created"
    assert len(cx.index) == len(o)
    gone = pre[3]
    gone.deleted = True                                       # tombstoned in storage, still indexed (quirk Q2): search skips it
    for q in ["authentication token", "database vacuum", "wavefront", "garlic pasta", "lifetime rust"]:
        got = cx.search(q, 5)
        exp = [r for r in o.search(emb.embed(q), 5) if st.get_node(uuid.UUID(bytes=bytes(r["node_id"]))) is not None]
        assert [n.id.bytes for _, n in got] == [bytes(r["node_id"]) for r in exp], q
        assert np.allclose([s for s, _ in got], [float(r["score"]) for r in exp], atol=SCORE_TOL)
        assert all(n.id != gone.id for _, n in got)
    assert cx.search("anything", 0) == []
    with pytest.raises(hip.ValidationError):                  # api.rs:62: a wrong-length stored embedding fails open()
        bad = MemStorage()
        bad.put_node(Node(uuid.uuid4(), "fact", embedding=np.zeros(DIM + 1, np.float32)))
        Cortex.open(bad, emb, devices)
