"""cx_search_decayed against the reference handler's sequence restated in Python
(cortex-server/src/http/routes.rs:889-947 via oracle/scoring.py): same ids, same order, same scores."""
import uuid

import numpy as np
import pytest

import bincode_ref as B
from conftest import ids_for

pytestmark = pytest.mark.gpu
NOW = (1_760_000_000, 250_000_000)
KINDS = ["event", "observation", "decision", "pattern", "fact", "preference", "note"]


def _corpus(hip, oracle, n, d, seed):
    rng = np.random.default_rng(seed)
    rows = oracle.synth_rows(n, d)
    ids = ids_for(n)
    h = hip.HipIndex(d)
    h.insert_batch(ids, rows)
    nodes = {}
    for i in range(n):
        la = (NOW[0] - int(rng.integers(0, 500 * 86400)), int(rng.integers(0, 1_000_000_000)))
        nodes[ids[i].tobytes()] = (KINDS[int(rng.integers(0, len(KINDS)))], la, int(rng.choice([0, 0, 1, 5, 30, 1000])))
    return h, rows, ids, nodes


def _set_stats(h, ids, nodes):
    keys = [ids[i].tobytes() for i in range(len(ids))]
    h.set_node_stats(ids, [nodes[k][0] for k in keys], [nodes[k][1] for k in keys], [nodes[k][2] for k in keys])


@pytest.mark.parametrize("limit,rb", [(10, None), (5, 0.5), (20, 1.0), (10, 0.0), (3, 0.01)])
def test_matches_the_handler_sequence(hip, oracle, limit, rb):
    from cortex_amd import scoring as S
    from oracle import scoring as O
    n, d = 4000, 384
    h, rows, ids, nodes = _corpus(hip, oracle, n, d, seed=21)
    _set_stats(h, ids, nodes)
    cfg, ocfg = S.ScoreDecayConfig(), O.ScoreDecayConfig()
    rbv = cfg.recency_weight if rb is None else rb
    qs = oracle.synth_queries(n, d, 6)
    for q in qs:
        got = h.search_decayed(q, limit, cfg, recency_bias=rb, now=NOW)
        cand = h.search(q, O.http_candidate_limit(limit, ocfg, rbv), None)     # the index's own results, as the handler sees them
        want = O.rerank([(r.node_id.bytes, r.score) for r in cand], nodes, limit, ocfg, rbv, NOW)
        assert [g[0].bytes for g in got] == [w[0] for w in want]
        assert [np.float32(g[1]).tobytes() for g in got] == [np.float32(w[1]).tobytes() for w in want]
        assert [np.float32(g[2]).tobytes() for g in got] == [np.float32(w[2]).tobytes() for w in want]
        if rbv > 0.0:
            assert len(got) == limit and any(g[1] != g[2] for g in got)
        else:
            assert [g[0] for g in got] == [r.node_id for r in cand[:limit]]    # no decay: plain search order


def test_disabled_config_and_filter_and_defaults(hip, oracle):
    from cortex_amd import scoring as S
    from oracle import scoring as O
    n, d = 1500, 384
    h, rows, ids, nodes = _corpus(hip, oracle, n, d, seed=22)
    q = oracle.synth_queries(n, d, 1)[0]
    # no stats set: every node reads as never accessed since the epoch -> the floor factor for all, order unchanged
    cfg = S.ScoreDecayConfig()
    got = h.search_decayed(q, 10, cfg, now=NOW)
    plain = h.search(q, 30, None)
    assert [g[0] for g in got] == [r.node_id for r in plain[:10]]
    floor = [float(O.apply_score_decay("", (0, 0), 0, r.score, O.ScoreDecayConfig(), 0.15, NOW)) for r in plain[:10]]
    assert [g[1] for g in got] == floor
    # disabled: raw scores, `limit` candidates
    got = h.search_decayed(q, 10, S.ScoreDecayConfig(enabled=False), now=NOW)
    assert [(g[0], g[1]) for g in got] == [(r.node_id, r.score) for r in plain[:10]]
    # filter is applied to the candidate search
    _set_stats(h, ids, nodes)
    ex = [plain[0].node_id.bytes, plain[1].node_id.bytes]
    got = h.search_decayed(q, 10, cfg, now=NOW, filter=hip.VectorFilter(exclude=ex))
    assert all(g[0].bytes not in ex for g in got) and len(got) == 10


def test_stats_survive_rebuild_and_come_from_bulk_load(hip, oracle):
    from cortex_amd import scoring as S
    from oracle import scoring as O
    n, d = 600, 384
    rows = oracle.synth_rows(n, d)
    rng = np.random.default_rng(23)
    recs, nodes = [], {}
    for i in range(n):
        i16 = uuid.UUID(int=(int(rng.integers(1, 1 << 62)) << 32) | i).bytes
        kind = KINDS[i % len(KINDS)]
        la_s = NOW[0] - int(rng.integers(0, 300 * 86400))
        import datetime
        la = datetime.datetime.fromtimestamp(la_s, datetime.timezone.utc).strftime("%Y-%m-%dT%H:%M:%SZ")
        ac = int(rng.integers(0, 50))
        nodes[i16] = (kind, (la_s, 0), ac)
        recs.append(B.encode_node(i16, kind, "t", "b", [], rows[i], "kai", None, None, 0.5, ac, la,
                                  "2024-01-01T00:00:%02dZ" % (i % 60), "2024-01-01T00:00:00Z", False))
    h = hip.HipIndex(d)
    st = h.bulk_load_nodes(recs, set_stats=True)
    assert st["indexed"] == n
    cfg, ocfg = S.ScoreDecayConfig(), O.ScoreDecayConfig()
    q = oracle.synth_queries(n, d, 1)[0]

    def check():
        got = h.search_decayed(q, 10, cfg, now=NOW)
        cand = h.search(q, 30, None)
        want = O.rerank([(r.node_id.bytes, r.score) for r in cand], nodes, 10, ocfg, cfg.recency_weight, NOW)
        assert [(g[0].bytes, np.float32(g[1]).tobytes()) for g in got] == [(w[0], np.float32(w[1]).tobytes()) for w in want]

    check()
    victims = [h.row_id(r) for r in range(0, 200, 3)]
    for v in victims:
        h.remove(v)
        nodes.pop(v.bytes)
    h.rebuild()            # rows move; the stats move with them
    check()
