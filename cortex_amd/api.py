"""Host-side mirror of the embedded facade `Cortex::{open, store, search}` (crates/cortex-core/src/api.rs:50-131) over
the HIP index — the L4 call sites of SURVEY §8 a18.

Everything that is not the vector layer is INJECTED and stays the reference's: `storage` is the redb-backed store
(`list_nodes()`, `put_node(node)`, `get_node(id)`), `embedding` the text-embedding service (`dimension()`,
`embed(text)`: FastEmbed/ONNX in the reference, out of scope here — the engine starts at "vector in hand").  What this
class pins is the SEQUENCE of index calls the facade makes, so that swapping `HnswIndex` for `HipIndex` /
`ShardedHipIndex` at api.rs:41,57 can be checked without a Rust toolchain:

  open   api.rs:56-70   insert every stored embedding in `list_nodes` order, `rebuild()` if there was any
  store  api.rs:99-114  embed `embedding_input(node)` if the node has no embedding, put_node, insert, notify hooks
  search api.rs:117-131 embed the query, `search(&emb, limit, None)`, hydrate each hit, skip ids storage does not have
"""
from __future__ import annotations

import uuid
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from .index import HipIndex, ShardedHipIndex


@dataclass
class Node:
    """The fields of `Node` (types.rs:26-68) the facade and the linker read."""
    id: uuid.UUID
    kind: str
    title: str = ""
    body: str = ""
    tags: List[str] = field(default_factory=list)
    source_agent: str = ""
    embedding: Optional[np.ndarray] = None
    deleted: bool = False


def embedding_input(node: Node) -> str:
    """vector/embedding.rs:113-131: "Kind: title\\nbody\\ntags: a, b" with the kind's first letter upper-cased."""
    k = node.kind
    kind_display = (k[0].upper() + k[1:]) if k else ""
    return f"{kind_display}: {node.title}\n{node.body}\ntags: {', '.join(node.tags)}"


class Cortex:
    def __init__(self, storage, embedding, index):
        self.storage, self.embedding, self.index = storage, embedding, index
        self.hooks: List[Callable[[Node, str], None]] = []

    @classmethod
    def open(cls, storage, embedding, devices: Optional[Sequence[int]] = None, dtype: str = "f32") -> "Cortex":
        """api.rs:50-82.  devices: None / one device -> HipIndex; several -> ShardedHipIndex (one shard per device)."""
        dim = embedding.dimension()
        # dtype "bf16": a bf16 row store (cx_create_ex) — half the HBM per embedding, results for the rounded vectors
        idx = HipIndex(dim, devices[0] if devices else 0, dtype=dtype) if not devices or len(devices) == 1 else ShardedHipIndex(dim, devices, dtype=dtype)
        any_ = False
        for node in storage.list_nodes():                      # NodeFilter::new(): deleted nodes are not listed
            if node.embedding is not None:
                idx.insert(node.id, node.embedding)            # a wrong-length embedding is an error here (the `?` at :62)
                any_ = True
        if any_:
            idx.rebuild()
        return cls(storage, embedding, idx)

    def add_hook(self, hook: Callable[[Node, str], None]) -> None:   # api.rs:85-87
        self.hooks.append(hook)

    def store(self, node: Node) -> uuid.UUID:
        """api.rs:99-114."""
        if node.embedding is None:
            node.embedding = np.asarray(self.embedding.embed(embedding_input(node)), dtype=np.float32)
        self.storage.put_node(node)
        self.index.insert(node.id, node.embedding)
        for h in self.hooks:
            h(node, "Created")
        return node.id

    def search(self, query: str, limit: int) -> List[Tuple[float, Node]]:
        """api.rs:117-131."""
        q = np.asarray(self.embedding.embed(query), dtype=np.float32)
        out = []
        for r in self.index.search(q, limit):
            node = self.storage.get_node(r.node_id)
            if node is not None:
                out.append((r.score, node))
        return out

    def get_node(self, node_id: uuid.UUID) -> Optional[Node]:       # api.rs:134-136
        return self.storage.get_node(node_id)
