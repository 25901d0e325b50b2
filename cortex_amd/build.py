"""Build libcortex_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "lib", "libcortex_hip.so")


def build(force: bool = False, jobs: int = 8) -> str:
    cmd = ["make", "-C", CSRC, f"-j{jobs}", "all"]   # the product library + the tests' fault-injection build of it
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"build did not produce {LIB_PATH}")
    return LIB_PATH


if __name__ == "__main__":
    print(build())
