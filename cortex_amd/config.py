"""SimilarityConfig — crates/cortex-core/src/vector/config.rs:3-87, kept verbatim in meaning."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .index import ValidationError


def _clamp01(x: float) -> float:
    return float(np.clip(np.float32(x), np.float32(0.0), np.float32(1.0)))


@dataclass
class SimilarityConfig:
    auto_link_threshold: float = float(np.float32(0.75))
    dedup_threshold: float = float(np.float32(0.92))
    contradiction_threshold: float = float(np.float32(0.80))
    auto_link_k: int = 20  # dead in the reference: the linker hard-codes 100 (auto_linker.rs:221)

    @staticmethod
    def new() -> "SimilarityConfig":
        return SimilarityConfig()

    @staticmethod
    def default() -> "SimilarityConfig":
        return SimilarityConfig()

    def with_auto_link_threshold(self, t: float) -> "SimilarityConfig":
        self.auto_link_threshold = _clamp01(t)
        return self

    def with_dedup_threshold(self, t: float) -> "SimilarityConfig":
        self.dedup_threshold = _clamp01(t)
        return self

    def with_contradiction_threshold(self, t: float) -> "SimilarityConfig":
        self.contradiction_threshold = _clamp01(t)
        return self

    def with_auto_link_k(self, k: int) -> "SimilarityConfig":
        self.auto_link_k = int(k)
        return self

    def validate(self) -> None:
        """vector/config.rs:66-87 — same checks, same order, same messages."""
        if np.float32(self.auto_link_threshold) >= np.float32(self.dedup_threshold):
            raise ValidationError("auto_link_threshold must be less than dedup_threshold")
        if np.float32(self.contradiction_threshold) >= np.float32(self.dedup_threshold):
            raise ValidationError("contradiction_threshold must be less than dedup_threshold")
        if self.auto_link_k == 0:
            raise ValidationError("auto_link_k must be greater than 0")
