"""Query-time score decay: the oracle's restatement (oracle/scoring.py) against the reference's own tests
(crates/cortex-core/src/vector/scoring.rs:134-262, re-expressed with `now` as an argument), and the C ABI's
cx_apply_score_decay against the oracle, bit for bit.  Host only."""
import numpy as np
import pytest

from cortex_amd import scoring as S
from oracle import scoring as O

NOW = (1_760_000_000, 123_456_789)
DAY = 86_400


def ago(days, ns=0):
    return (NOW[0] - days * DAY, ns)


@pytest.fixture(params=["oracle", "abi"])
def decay(request):
    if request.param == "oracle":
        return lambda kind, la, ac, raw, cfg, rb: float(O.apply_score_decay(kind, la, ac, raw, cfg, rb, NOW)), O.ScoreDecayConfig
    return lambda kind, la, ac, raw, cfg, rb: S.apply_score_decay(kind, la, ac, raw, cfg, rb, NOW), S.ScoreDecayConfig


def test_decay_disabled_returns_raw(decay):            # scoring.rs:136-144
    f, Cfg = decay
    assert f("fact", NOW, 0, 0.8, Cfg(enabled=False), 0.15) == float(np.float32(0.8))


def test_zero_recency_bias_returns_raw(decay):         # :146-151
    f, Cfg = decay
    assert f("fact", NOW, 0, 0.8, Cfg(), 0.0) == float(np.float32(0.8))


def test_fresh_node_no_decay(decay):                   # :153-166
    f, Cfg = decay
    cfg = Cfg()
    assert abs(f("fact", NOW, 0, 0.8, cfg, cfg.recency_weight) - 0.8) < 0.01


def test_stale_node_decays(decay):                     # :168-186
    f, Cfg = decay
    cfg = Cfg()
    assert f("fact", ago(100), 0, 0.8, cfg, cfg.recency_weight) < 0.8


def test_floor_at_min_factor(decay):                   # :188-209
    f, Cfg = decay
    cfg = Cfg()
    floor_score = 0.8 * (1.0 - cfg.recency_weight) + 0.8 * cfg.min_factor * 1.0 * cfg.recency_weight
    assert abs(f("event", ago(400), 0, 0.8, cfg, cfg.recency_weight) - floor_score) < 0.01


def test_echo_boost_capped(decay):                     # :211-229
    f, Cfg = decay
    cfg = Cfg()
    expected = 0.8 * (1.0 - cfg.recency_weight) + 0.8 * 1.0 * cfg.echo_cap * cfg.recency_weight
    assert abs(f("fact", NOW, 10_000, 0.8, cfg, cfg.recency_weight) - expected) < 0.01


def test_kind_rate_override(decay):                    # :231-251
    f, Cfg = decay
    cfg = Cfg()
    assert f("decision", ago(30), 0, 0.8, cfg, cfg.recency_weight) > f("event", ago(30), 0, 0.8, cfg, cfg.recency_weight)


def test_recency_bias_zero_equals_raw(decay):          # :253-262
    f, Cfg = decay
    assert f("fact", ago(200), 5, 0.75, Cfg(), 0.0) == 0.75


def test_recency_bias_one_full_decay(decay):           # :264-276
    f, Cfg = decay
    assert abs(f("fact", NOW, 0, 0.9, Cfg(), 1.0) - 0.9) < 0.01


def test_abi_is_bit_identical_to_the_oracle():
    rng = np.random.default_rng(7)
    kinds = ["event", "observation", "decision", "pattern", "fact", "preference", "other-kind"]
    ocfg, scfg = O.ScoreDecayConfig(), S.ScoreDecayConfig()
    for _ in range(3000):
        kind = kinds[int(rng.integers(0, len(kinds)))]
        la = (NOW[0] - int(rng.integers(-10, 500 * DAY)), int(rng.integers(0, 1_000_000_000)))
        ac = int(rng.choice([0, 1, 3, 19, 20, 21, 10_000]))
        raw = float(np.float32(rng.random()))
        rb = float(np.float32(rng.choice([0.15, 0.5, 1.0, 0.01])))
        a = O.apply_score_decay(kind, la, ac, raw, ocfg, rb, NOW)
        b = S.apply_score_decay(kind, la, ac, raw, scfg, rb, NOW)
        assert np.float32(b).tobytes() == np.float32(a).tobytes(), (kind, la, ac, raw, rb, a, b)


def test_num_seconds_truncates_toward_zero():
    assert O.num_seconds((10, 0), (8, 500_000_000)) == 1      # 1.5 s
    assert O.num_seconds((10, 0), (11, 500_000_000)) == -1    # -1.5 s -> -1, then .max(0)
    # an access "in the future" clamps to zero days idle: score as fresh
    cfg = O.ScoreDecayConfig()
    assert O.apply_score_decay("fact", (NOW[0] + 5, 0), 0, 0.8, cfg, 0.15, NOW) == O.apply_score_decay("fact", NOW, 0, 0.8, cfg, 0.15, NOW)
    assert S.apply_score_decay("fact", (NOW[0] + 5, 0), 0, 0.8, S.ScoreDecayConfig(), 0.15, NOW) == float(O.apply_score_decay("fact", NOW, 0, 0.8, cfg, 0.15, NOW))


def test_http_candidate_limit():                        # routes.rs:899-903
    cfg = S.ScoreDecayConfig()
    assert S.http_candidate_limit(10, cfg, 0.15) == 30 and S.http_candidate_limit(20, cfg, 0.15) == 60
    assert S.http_candidate_limit(10, cfg, 0.0) == 10 and S.http_candidate_limit(10, S.ScoreDecayConfig(enabled=False), 0.5) == 10
