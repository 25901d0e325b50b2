"""CPU restatement (test infrastructure only) of the reference's query-time score decay and of the re-rank its HTTP
search handler does with it.  Follows crates/cortex-core/src/vector/scoring.rs:22-114 (ScoreDecayConfig,
apply_score_decay) and crates/cortex-server/src/http/routes.rs:889-947 (candidate_limit, decay, stable sort,
truncate).  Pinned by the reference's own tests, scoring.rs:134-262 (tests/test_scoring_kat.py).

f64 where the reference computes in f64 (math.exp is the C library's exp, like Rust's f64::exp), numpy f32 with
one rounding per operation where it computes in f32."""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

import numpy as np

f32 = np.float32


def _default_by_kind() -> Dict[str, float]:   # scoring.rs:59-66
    return {"event": 0.05, "observation": 0.04, "decision": 0.005, "pattern": 0.005, "fact": 0.01, "preference": 0.005}


@dataclass
class ScoreDecayConfig:   # scoring.rs:22-78
    enabled: bool = True
    daily_rate: float = 0.02
    max_age_days: float = 365.0
    min_factor: float = 0.1
    echo_weight: float = 0.05
    echo_cap: float = 2.0
    recency_weight: float = 0.15
    by_kind: Dict[str, float] = field(default_factory=_default_by_kind)


def num_seconds(now: Tuple[int, int], then: Tuple[int, int]) -> int:
    """chrono Duration::num_seconds of now - then: whole seconds, truncated toward zero; times are (s, ns)."""
    total_ns = (now[0] - then[0]) * 1_000_000_000 + (now[1] - then[1])
    return int(total_ns / 1_000_000_000) if abs(total_ns) < (1 << 52) else (abs(total_ns) // 1_000_000_000) * (1 if total_ns >= 0 else -1)


def apply_score_decay(kind: str, last_accessed_at: Tuple[int, int], access_count: int, raw_score, config: ScoreDecayConfig,
                      recency_bias, now: Tuple[int, int]) -> np.float32:
    """scoring.rs:84-114 with `now` as an argument."""
    raw, rb = f32(raw_score), f32(recency_bias)
    if not config.enabled or rb == f32(0.0):
        return raw
    days_idle = float(max(num_seconds(now, last_accessed_at), 0)) / 86400.0
    kind_rate = config.by_kind.get(kind, config.daily_rate)
    effective_days = min(days_idle, config.max_age_days)
    temporal = f32(max(math.exp(-kind_rate * effective_days), config.min_factor))
    echo = f32(min(1.0 + float(access_count) * config.echo_weight, config.echo_cap))
    a = f32(raw * f32(f32(1.0) - rb))
    b = f32(f32(f32(raw * temporal) * echo) * rb)
    return f32(a + b)


def http_candidate_limit(limit: int, config: ScoreDecayConfig, recency_bias: float) -> int:
    """routes.rs:899-903"""
    return max(limit * 3, 30) if config.enabled and recency_bias > 0.0 else limit


def rerank(results: Sequence[Tuple[bytes, float]], nodes: Dict[bytes, Tuple[str, Tuple[int, int], int]], limit: int,
           config: ScoreDecayConfig, recency_bias: float, now: Tuple[int, int]) -> List[Tuple[bytes, np.float32, np.float32]]:
    """routes.rs:909-947: results = [(id, raw score)] in search order; nodes[id] = (kind, last_accessed_at, access_count);
    ids missing from `nodes` are dropped like the handler's filter_map.  -> [(id, final, raw)] of length <= limit."""
    scored = []
    for i, raw in results:
        if i not in nodes:
            continue
        kind, la, ac = nodes[i]
        scored.append((i, apply_score_decay(kind, la, ac, raw, config, recency_bias, now), f32(raw)))
    # sort_by(|a, b| b.1.partial_cmp(&a.1).unwrap_or(Equal)): stable, descending
    order = sorted(range(len(scored)), key=lambda j: -float(scored[j][1]) if scored[j][1] == scored[j][1] else 0.0)
    return [scored[j] for j in order][:limit]
