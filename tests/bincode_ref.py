"""Independent Python statement of bincode 1.3's default encoding for the two layouts the hot path
touches (test infrastructure): the index file of HnswIndex::save (vector/index.rs:437-445) and the
stored `Node` (types.rs:26-68), whose golden bytes the reference pins in storage/redb_storage.rs:1827-1857.
Little endian, fixed-width ints, u64 lengths; Uuid = serialize_bytes (u64 16 + 16 bytes);
Option = 1 tag byte; DateTime<Utc> = RFC 3339 string."""
import struct
from typing import Dict, List, Optional, Tuple

import numpy as np


class R:
    def __init__(self, b: bytes):
        self.b, self.o = b, 0

    def take(self, n: int) -> bytes:
        assert self.o + n <= len(self.b), "truncated"
        v = self.b[self.o:self.o + n]
        self.o += n
        return v

    def u8(self) -> int: return self.take(1)[0]
    def u64(self) -> int: return struct.unpack("<Q", self.take(8))[0]
    def f32(self) -> float: return struct.unpack("<f", self.take(4))[0]
    def string(self) -> str: return self.take(self.u64()).decode()
    def uuid(self) -> bytes:
        assert self.u64() == 16
        return self.take(16)
    def vec_f32(self) -> np.ndarray:
        n = self.u64()
        return np.frombuffer(self.take(4 * n), dtype="<f4").copy()
    def opt(self, f):
        t = self.u8()
        assert t in (0, 1)
        return f() if t else None


def decode_index_file(b: bytes):
    """-> (vectors: [(id16, f32[])] in file order, metadata: {id16: (kind, agent)}, dimension)"""
    r = R(b)
    vectors = [(r.uuid(), r.vec_f32()) for _ in range(r.u64())]
    meta = {}
    for _ in range(r.u64()):
        i = r.uuid()
        meta[i] = (r.string(), r.string())
    dim = r.u64()
    assert r.o == len(b), "trailing bytes"
    return vectors, meta, dim


def encode_index_file(vectors: List[Tuple[bytes, np.ndarray]], meta: Dict[bytes, Tuple[str, str]], dim: int) -> bytes:
    out = [struct.pack("<Q", len(vectors))]
    for i, v in vectors:
        v = np.asarray(v, dtype="<f4")
        out += [struct.pack("<Q", 16), i, struct.pack("<Q", len(v)), v.tobytes()]
    out.append(struct.pack("<Q", len(meta)))
    for i, (k, a) in meta.items():
        kb, ab = k.encode(), a.encode()
        out += [struct.pack("<Q", 16), i, struct.pack("<Q", len(kb)), kb, struct.pack("<Q", len(ab)), ab]
    out.append(struct.pack("<Q", dim))
    return b"".join(out)


def decode_node(b: bytes) -> dict:
    """Node with an EMPTY data.metadata map (serde_json::Value has no bincode decoding)."""
    r = R(b)
    n = {"id": r.uuid(), "kind": r.string(), "title": r.string(), "body": r.string()}
    assert r.u64() == 0, "non-empty NodeData.metadata is not decodable"
    n["tags"] = [r.string() for _ in range(r.u64())]
    n["embedding"] = r.opt(r.vec_f32)
    n["agent"] = r.string()
    n["session"] = r.opt(r.string)
    n["channel"] = r.opt(r.string)
    n["importance"] = r.f32()
    n["access_count"] = r.u64()
    n["last_accessed_at"], n["created_at"], n["updated_at"] = r.string(), r.string(), r.string()
    d = r.u8()
    assert d in (0, 1), "invalid bool encoding"
    n["deleted"] = bool(d)
    assert r.o == len(b)
    return n


def _s(x: str) -> bytes:
    b = x.encode()
    return struct.pack("<Q", len(b)) + b


def _opt_s(x: Optional[str]) -> bytes:
    return b"\x00" if x is None else b"\x01" + _s(x)


def encode_node(id16: bytes, kind: str, title: str, body: str, tags: List[str], embedding, agent: str,
                session: Optional[str], channel: Optional[str], importance: float, access_count: int,
                last_accessed_at: str, created_at: str, updated_at: str, deleted: bool) -> bytes:
    """bincode of `Node` (types.rs:26-68) with an empty data.metadata map; the inverse of decode_node, pinned by
    re-encoding the reference's golden bytes (test_persist_format.py)."""
    out = [struct.pack("<Q", 16), id16, _s(kind), _s(title), _s(body), struct.pack("<Q", 0),
           struct.pack("<Q", len(tags))] + [_s(t) for t in tags]
    if embedding is None:
        out.append(b"\x00")
    else:
        e = np.asarray(embedding, dtype="<f4")
        out += [b"\x01", struct.pack("<Q", e.size), e.tobytes()]
    out += [_s(agent), _opt_s(session), _opt_s(channel), struct.pack("<f", importance), struct.pack("<Q", access_count),
            _s(last_accessed_at), _s(created_at), _s(updated_at), b"\x01" if deleted else b"\x00"]
    return b"".join(out)
