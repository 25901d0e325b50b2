"""Query-time score decay (vector/scoring.rs:22-114) and the HTTP handler's re-rank (routes.rs:889-947) over the
C ABI (cx_apply_score_decay, cx_search_decayed).  Same names and argument meaning as the reference; `now` is an
argument (seconds, nanoseconds since the epoch) where the reference reads Utc::now()."""
from __future__ import annotations

import ctypes as C
import time
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import numpy as np

from . import _lib


def _default_by_kind() -> Dict[str, float]:   # scoring.rs:59-66
    return {"event": 0.05, "observation": 0.04, "decision": 0.005, "pattern": 0.005, "fact": 0.01, "preference": 0.005}


@dataclass
class ScoreDecayConfig:   # scoring.rs:22-78, Default :57-78
    enabled: bool = True
    daily_rate: float = 0.02
    max_age_days: float = 365.0
    min_factor: float = 0.1
    echo_weight: float = 0.05
    echo_cap: float = 2.0
    recency_weight: float = 0.15
    by_kind: Dict[str, float] = field(default_factory=_default_by_kind)

    def _c(self, intern):
        """-> (cx_decay_config, keep-alive arrays); intern: kind string -> code"""
        codes = np.asarray([intern(k) for k in self.by_kind], np.uint32)
        rates = np.asarray(list(self.by_kind.values()), np.float64)
        c = _lib.cx_decay_config(int(self.enabled), self.daily_rate, self.max_age_days, self.min_factor, self.echo_weight,
                                 self.echo_cap, self.recency_weight, len(codes),
                                 codes.ctypes.data if len(codes) else None, rates.ctypes.data if len(rates) else None)
        return c, (codes, rates)


def now_utc() -> Tuple[int, int]:
    t = time.time_ns()
    return t // 1_000_000_000, t % 1_000_000_000


def http_candidate_limit(limit: int, config: ScoreDecayConfig, recency_bias: float) -> int:
    """routes.rs:899-903: extra candidates so the re-rank does not cut off fresher nodes."""
    return max(limit * 3, 30) if config.enabled and recency_bias > 0.0 else limit


def apply_score_decay(kind: str, last_accessed_at: Tuple[int, int], access_count: int, raw_score: float,
                      config: ScoreDecayConfig, recency_bias: float, now: Optional[Tuple[int, int]] = None) -> float:
    """scoring.rs:84-114 for a node given by the three fields the formula reads."""
    L = _lib.load()
    names = list(config.by_kind)
    code = {k: i + 1 for i, k in enumerate(names)}   # any injective coding does: only equality is used
    c, keep = config._c(lambda k: code[k])
    now = now_utc() if now is None else now
    return float(L.cx_apply_score_decay(C.byref(c), raw_score, recency_bias, now[0], now[1], code.get(kind, 0),
                                        last_accessed_at[0], last_accessed_at[1], access_count))
