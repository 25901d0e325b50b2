"""cortex_amd — MI355X-native similarity engine behind cortex-core's VectorIndex seam.

The product is libcortex_hip.so (cortex_amd/csrc, C ABI in include/cortex_hip.h);
this package is the host-side mirror of the reference's interface for that path.
"""
from .config import SimilarityConfig
from .index import CortexError, HipIndex, ShardedHipIndex, SimilarityResult, ValidationError, VectorFilter

__all__ = ["HipIndex", "ShardedHipIndex", "VectorFilter", "SimilarityResult", "SimilarityConfig", "CortexError", "ValidationError"]
