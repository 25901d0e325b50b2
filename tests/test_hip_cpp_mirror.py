"""Runs tests/cpp/test_vector_index.cpp — the reference's vector-layer tests against the C++ host
mirror (include/cortex_hip.hpp) — built by __graft_entry__.build()."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "test_vector_index")


def test_cpp_mirror_compiles_here():
    """not gpu: the header-only mirror and its test build with plain g++ against the C ABI."""
    import __graft_entry__ as g
    g.build_cpp_tests()
    assert os.path.exists(BIN)


@pytest.mark.gpu
def test_reference_tests_pass_through_the_cpp_mirror(tmp_path):
    if not os.path.exists(BIN):
        import __graft_entry__ as g
        g.build_cpp_tests()
    env = dict(os.environ, TMPDIR=str(tmp_path))
    p = subprocess.run([BIN], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "18 tests run, 0 checks failed" in p.stdout
