"""The bincode helper used by the persistence tests is pinned by the reference's own golden Node bytes
(storage/redb_storage.rs:1827-1857): Uuid = u64 16 + raw bytes, String = u64 len + UTF-8, Option tag
byte, f32/u64 little endian — the same primitives the index file (vector/index.rs:437-473) is made of."""
import json
import os

import numpy as np

import bincode_ref as B

HERE = os.path.dirname(__file__)


def test_golden_node_bytes_decode():
    g = json.load(open(os.path.join(HERE, "golden", "node_schema_golden.json")))
    n = B.decode_node(bytes(g["bytes"]))
    e = g["expect"]
    assert n["id"].hex() == e["id_hex"]
    for k in ("kind", "title", "body", "tags", "embedding", "agent", "session", "channel", "access_count",
              "last_accessed_at", "created_at", "updated_at", "deleted"):
        assert n[k] == e[k], k
    assert n["importance"] == e["importance"]


def test_index_file_roundtrip_in_python():
    rng = np.random.default_rng(0)
    vecs = [(bytes(rng.integers(0, 256, 16, dtype=np.uint8)), rng.standard_normal(5).astype(np.float32)) for _ in range(7)]
    meta = {vecs[1][0]: ("fact", "kai"), vecs[4][0]: ("decision", "")}
    b = B.encode_index_file(vecs, meta, 5)
    v2, m2, d2 = B.decode_index_file(b)
    assert d2 == 5 and m2 == meta and all(a[0] == c[0] and np.array_equal(a[1], c[1]) for a, c in zip(vecs, v2))
    # layout spot checks: leading map length, first key is a length-prefixed 16-byte uuid
    assert b[:8] == (7).to_bytes(8, "little") and b[8:16] == (16).to_bytes(8, "little") and b[16:32] == vecs[0][0]
    assert b[-8:] == (5).to_bytes(8, "little")
