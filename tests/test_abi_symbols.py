"""The C-ABI library loads here (no GPU) and exports every symbol include/cortex_hip.h declares;
no compute call works without a device and none falls back to the CPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for hdr in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if not hdr.endswith(".h"):
            continue
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(cx_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    from cortex_amd import _lib
    L = _lib.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/ but not exported"
    assert set(names) == set(_lib.SIGNATURES), "ctypes table and header disagree"


def test_no_cpu_fallback_without_device():
    from cortex_amd import _lib
    import cortex_amd
    L = _lib.load()
    if L.cx_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(cortex_amd.CortexError, match="no CPU fallback"):
        cortex_amd.HipIndex(3)


def test_product_never_imports_oracle():
    """cortex_amd/ (the product) must not reference oracle/ in any way."""
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "cortex_amd")):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")) or f == "Makefile":
                t = open(os.path.join(dp, f), errors="replace").read()
                if re.search(r"\boracle\b|cxo_|cxs_", t):
                    bad.append(os.path.join(dp, f))
    assert not bad, f"product files mention the oracle: {bad}"


def test_similarity_config_mirror():  # vector/config.rs:93-135
    import numpy as np
    from cortex_amd import SimilarityConfig, ValidationError
    c = SimilarityConfig.default()
    assert (np.float32(c.auto_link_threshold), np.float32(c.dedup_threshold),
            np.float32(c.contradiction_threshold), c.auto_link_k) == (np.float32(0.75), np.float32(0.92), np.float32(0.80), 20)
    c.validate()
    c = SimilarityConfig.new().with_auto_link_threshold(0.70).with_dedup_threshold(0.95).with_auto_link_k(30)
    assert (np.float32(c.auto_link_threshold), np.float32(c.dedup_threshold), c.auto_link_k) == (np.float32(0.70), np.float32(0.95), 30)
    bad = SimilarityConfig.new().with_auto_link_threshold(0.95).with_dedup_threshold(0.90)
    with pytest.raises(ValidationError):
        bad.validate()
    c = SimilarityConfig.new().with_auto_link_threshold(1.5).with_dedup_threshold(-0.5)
    assert c.auto_link_threshold == 1.0 and c.dedup_threshold == 0.0


def test_poisoned_result_block_is_an_error_not_an_index():
    """Every host entry point checks a result block it read back from the device before using it (a
    kernel bug once became a SIGSEGV in the caller).  The guard itself, fed a poisoned block: no GPU needed."""
    import ctypes as C
    import numpy as np
    from cortex_amd import _lib
    L = _lib.load()
    nq, k, n_rows = 3, 4, 100
    counts = np.array([4, 2, 0], np.uint32)
    rows = np.zeros((nq, k), np.uint32)
    rows[0] = [5, 99, 0, 7]
    rows[1] = [1, 2, 0xFFFFFFFF, 0xFFFFFFFF]      # beyond counts[1]: never looked at
    chk = lambda: L.cx_debug_check_result_block(counts.ctypes.data, rows.ctypes.data, nq, k, k, n_rows)
    assert chk() == 0
    rows[0, 1] = 100                               # a row index one past the store
    assert chk() == 2 and b"row 100" in L.cx_last_error()
    rows[0, 1] = 99
    counts[2] = 5                                  # a list longer than k (the `lane < k` writer bug of round 1)
    assert chk() == 2 and b"5 entries" in L.cx_last_error()
    counts[2] = 0xFFFFFFFF
    assert chk() == 2


def test_storage_dtype_codes_match_the_header():
    """cx_create_ex's dtype codes: the header's CX_DTYPE_* against the ctypes mirror's table (cortex_amd/index.py)."""
    import re
    from cortex_amd.index import _dtype_code
    hdr = open(os.path.join(ROOT, "include", "cortex_hip.h")).read()
    codes = {m.group(1).lower(): int(m.group(2)) for m in re.finditer(r"#define\s+CX_DTYPE_(\w+)\s+(\d+)", hdr)}
    assert codes == {"f32": 0, "bf16": 1}
    for name, code in codes.items():
        assert _dtype_code(name) == code
    with pytest.raises(Exception):
        _dtype_code("fp8")


def test_fault_injection_is_not_in_the_product_library():
    """Round-3 ADVICE: CX_SHARD_FAIL_UPSERT was compiled into the production upsert path.  It now exists only in
    libcortex_hip_testhooks.so (csrc/Makefile: -DCX_TEST_HOOKS), which exports the same ABI; the product does not know the switch."""
    lib = os.path.join(ROOT, "cortex_amd", "lib", "libcortex_hip.so")
    hooks = os.path.join(ROOT, "cortex_amd", "lib", "libcortex_hip_testhooks.so")
    assert os.path.exists(lib) and os.path.exists(hooks), "__graft_entry__.build() makes both"
    assert b"CX_SHARD_FAIL_UPSERT" not in open(lib, "rb").read()
    assert b"CX_SHARD_FAIL_UPSERT" in open(hooks, "rb").read()
    H = ctypes.CDLL(hooks)
    for n in declared_symbols():
        assert hasattr(H, n), f"{n} missing from the test-hooks build"
