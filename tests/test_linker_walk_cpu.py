"""Host logic of the auto-linker mirror, without a GPU: `linker.autolink_walk` (the reference's per-node loop,
auto_linker.rs:215-264) driven by neighbour lists from the ORACLE instead of the HIP engine must reproduce the
oracle's own restatement of the similarity pass, and the cap / self / deleted / existing-edge rules."""
import uuid

import numpy as np
import pytest

from conftest import ids_for


class OracleBackedIndex:
    """the three methods autolink_walk needs, answered by the CPU oracle"""

    def __init__(self, oracle, rows, ids):
        self.o = oracle.OracleIndex(rows.shape[1])
        self.o.insert_batch(ids, rows)
        self.rows, self.ids = rows, ids

    def row_count(self):
        return len(self.rows)

    def row_id(self, r):
        return uuid.UUID(bytes=self.ids[r].tobytes())

    def topk_lists_rows(self, topk, scan_rows=None):
        scan = np.arange(len(self.rows)) if scan_rows is None else np.asarray(scan_rows)
        out_r = np.zeros((len(scan), topk), np.uint32)
        out_s = np.zeros((len(scan), topk), np.float32)
        cnt = np.zeros(len(scan), np.uint32)
        for p, i in enumerate(scan):
            e = self.o.search(self.rows[i], topk)
            cnt[p] = len(e)
            out_r[p, :len(e)] = e["row"]
            out_s[p, :len(e)] = e["score"]
        return out_r, out_s, cnt


@pytest.fixture(scope="module")
def setup():
    from oracle import oracle
    n, d = 600, 96
    rows = oracle.synth_rows(n, d)
    ids = ids_for(n)
    return oracle, rows, ids, OracleBackedIndex(oracle, rows, ids)


def test_walk_with_similarity_rule_equals_oracle_pass(setup):
    oracle, rows, ids, ix = setup
    from cortex_amd import linker
    from cortex_amd.config import SimilarityConfig
    cfg = SimilarityConfig(auto_link_threshold=0.7, dedup_threshold=0.95, contradiction_threshold=0.8)
    scan = np.array([0, 3, 17, 250, 599], dtype=np.uint32)
    scan_ids = [ids[i].tobytes() for i in scan]
    deleted = [ids[4].tobytes(), ids[251].tobytes()]
    walk = linker.autolink_walk(ix, scan_ids, [linker.similarity_rule(cfg)], max_edges_per_node=5, deleted_ids=deleted)
    flags = np.zeros(len(rows), np.uint8); flags[[4, 251]] = 1
    want = ix.o.autolink_pass(scan, 100, np.float32(0.7), 5, deleted=flags)
    assert [(a, b) for a, b, _, _ in walk] == [(int(e["from_row"]), int(e["to_row"])) for e in want]
    assert np.array_equal(np.array([w for *_, w in walk], np.float32), want["weight"])


def test_walk_cap_counts_every_rule_and_is_tested_per_neighbour(setup):
    oracle, rows, ids, ix = setup
    from cortex_amd import linker

    def two_edges(node, nb, score):          # fires twice for every neighbour
        return [("same_agent", 0.5), ("temporal", 0.25)]
    scan_ids = [ids[9].tobytes()]
    walk = linker.autolink_walk(ix, scan_ids, [two_edges], max_edges_per_node=5)
    # neighbours are walked in score order, self skipped; the cap is checked after each neighbour: 2, 4, 6 -> stop at 6
    assert len(walk) == 6
    nbs = [b for _, b, _, _ in walk]
    e = ix.o.search(rows[9], 100)
    order = [int(r) for r in e["row"] if int(r) != 9][:3]
    assert nbs == [order[0], order[0], order[1], order[1], order[2], order[2]]


def test_walk_skips_existing_edges_without_counting_them(setup):
    oracle, rows, ids, ix = setup
    from cortex_amd import linker
    e = ix.o.search(rows[20], 100)
    order = [int(r) for r in e["row"] if int(r) != 20]

    def always(node, nb, score):
        return [("related_to", score)]
    have = {(order[0], "related_to"), (order[2], "related_to")}
    walk = linker.autolink_walk(ix, [ids[20].tobytes()], [always], max_edges_per_node=3, existing=lambda node: have)
    assert [b for _, b, _, _ in walk] == [order[1], order[3], order[4]]


def _csr(lists):
    off = np.zeros(len(lists) + 1, np.uint64)
    off[1:] = np.cumsum([len(x) for x in lists])
    to = np.array([t for x in lists for t in x], dtype=np.uint32)
    return off, to


def test_oracle_pass_with_existing_edges_equals_the_walk(setup):
    """auto_linker.rs:226-231, :249-263, :284-287 in the oracle's pass: an edge the node already has is dropped
    WITHOUT counting towards max_edges_per_node (the walk goes deeper into the top-100 list), and the cycle keeps
    the first max_edges_per_cycle proposals.  Checked against the line-for-line Python walk over the oracle's lists."""
    oracle, rows, ids, ix = setup
    from cortex_amd import linker
    from cortex_amd.config import SimilarityConfig
    cfg = SimilarityConfig(auto_link_threshold=0.6, dedup_threshold=0.95, contradiction_threshold=0.8)
    rng = np.random.default_rng(3)
    scan = rng.permutation(len(rows))[:120].astype(np.uint32)
    first = ix.o.autolink_pass(scan, 100, np.float32(0.6), 4)            # a first cycle: 4 edges per node
    have = {int(s): [] for s in scan}
    for e in first:
        have[int(e["from_row"])].append(int(e["to_row"]))
    lists = []
    for p, s in enumerate(scan):
        if p % 3 == 0:
            lists.append(have[int(s)] + [int(x) for x in rng.integers(0, len(rows), 2)])   # its edges + unrelated ones
        elif p % 3 == 1:
            lists.append(have[int(s)][:2])                                                  # some of them
        else:
            lists.append([])
    off, to = _csr(lists)
    want = ix.o.autolink_pass(scan, 100, np.float32(0.6), 4, existing=(off, to))
    sets = {int(s): {(t, "related_to") for t in l} for s, l in zip(scan, lists)}
    walk = linker.autolink_walk(ix, [ids[i].tobytes() for i in scan], [linker.similarity_rule(cfg)], max_edges_per_node=4,
                                existing=lambda node: sets[node])
    assert [(a, b) for a, b, _, _ in walk] == [(int(e["from_row"]), int(e["to_row"])) for e in want]
    # nothing the node already has is proposed again, and nodes with all 4 first-cycle edges existing walk deeper
    for e in want:
        assert (int(e["to_row"]), "related_to") not in sets[int(e["from_row"])]
    deeper = [int(s) for p, s in enumerate(scan) if p % 3 == 0 and len(have[int(s)]) == 4]
    assert deeper and any(int(e["from_row"]) in deeper for e in want)
    # per-cycle truncation = the first max_edges_per_cycle proposals, in scan order
    cut = ix.o.autolink_pass(scan, 100, np.float32(0.6), 4, existing=(off, to), max_edges_per_cycle=37)
    assert len(cut) == 37 and np.array_equal(cut, want[:37])


def test_oracle_pass_cap_is_tested_after_the_push(setup):
    """:259-262 follows the push: with max_edges_per_node = 0 the first neighbour's edge still gets through."""
    oracle, rows, ids, ix = setup
    scan = np.array([9, 20], dtype=np.uint32)
    e0 = ix.o.autolink_pass(scan, 100, np.float32(0.0), 0)
    e1 = ix.o.autolink_pass(scan, 100, np.float32(0.0), 1)
    assert len(e0) == 2 and np.array_equal(e0, e1)
