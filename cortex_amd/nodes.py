"""Stored `Node` records (types.rs:26-68, bincode; golden bytes storage/redb_storage.rs:1827-1857) as the
bulk loader reads them — a thin view over `cx_node_decode` (include/cortex_hip.h).  Host only."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .index import ValidationError


def decode_node(record: bytes) -> dict:
    """RedbStorage::deserialize_node (redb_storage.rs:230-232) for the fields the vector layer and the
    linker read.  Raises ValidationError where the reference's deserialize fails (list_nodes skips those)."""
    L = _lib.load()
    buf = (C.c_uint8 * max(1, len(record))).from_buffer_copy(bytes(record) or b"\0")
    v = _lib.cx_node_view()
    if L.cx_node_decode(C.addressof(buf), len(record), C.byref(v)) != 0:
        raise ValidationError((L.cx_last_error() or b"").decode(errors="replace"))

    def s(p, n):
        return C.string_at(p, n).decode() if n else ""

    emb = None
    if v.has_embedding:
        emb = np.frombuffer(C.string_at(v.embedding, 4 * v.embedding_len), dtype="<f4").copy()
    return {
        "id": bytes(v.id), "kind": s(v.kind, v.kind_len), "title": s(v.title, v.title_len),
        "body": s(v.body, v.body_len), "n_tags": int(v.n_tags), "agent": s(v.agent, v.agent_len),
        "embedding": emb, "importance": float(v.importance), "access_count": int(v.access_count),
        "last_accessed_at": (int(v.last_accessed_at_s), int(v.last_accessed_at_ns)),
        "created_at": (int(v.created_at_s), int(v.created_at_ns)),
        "updated_at": (int(v.updated_at_s), int(v.updated_at_ns)),
        "deleted": bool(v.deleted), "bytes_used": int(v.bytes_used),
    }
