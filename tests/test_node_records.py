"""The bulk loader's record decoder (cx_node_decode, include/cortex_hip.h) against the reference's
golden `Node` bytes (storage/redb_storage.rs:1827-1857) and against the independent Python statement of the
same bincode layout (tests/bincode_ref.py).  Host only: no device is touched."""
import calendar
import json
import os
import struct

import numpy as np
import pytest

import bincode_ref as B
from cortex_amd.index import ValidationError
from cortex_amd.nodes import decode_node

HERE = os.path.dirname(__file__)
GOLD = json.load(open(os.path.join(HERE, "golden", "node_schema_golden.json")))


def _epoch(y, mo, d, h=0, mi=0, s=0):
    return calendar.timegm((y, mo, d, h, mi, s))


def test_reference_golden_bytes():
    rec = bytes(GOLD["bytes"])
    n, e = decode_node(rec), GOLD["expect"]
    assert n["id"].hex() == e["id_hex"]
    assert (n["kind"], n["title"], n["body"], n["agent"]) == (e["kind"], e["title"], e["body"], e["agent"])
    assert n["n_tags"] == len(e["tags"]) and n["embedding"] is None
    assert n["importance"] == e["importance"] == 0.5 and n["access_count"] == 0 and n["deleted"] is False
    # make_canonical_node (redb_storage.rs:1189-1191): UNIX_EPOCH and timestamp 1_700_000_000
    assert n["last_accessed_at"] == (0, 0) and n["created_at"] == (1_700_000_000, 0) == n["updated_at"]
    assert n["bytes_used"] == len(rec)


def test_encoder_reproduces_the_golden_bytes():
    e = GOLD["expect"]
    rec = B.encode_node(bytes.fromhex(e["id_hex"]), e["kind"], e["title"], e["body"], e["tags"], None, e["agent"],
                        e["session"], e["channel"], e["importance"], e["access_count"], e["last_accessed_at"],
                        e["created_at"], e["updated_at"], e["deleted"])
    assert rec == bytes(GOLD["bytes"])


def _node(rng, dim=7, **kw):
    d = dict(id16=bytes(rng.integers(0, 256, 16, dtype=np.uint8)), kind="fact", title="té ☃ \U0001F600", body="b" * int(rng.integers(0, 50)),
             tags=["a", "bb"], embedding=rng.standard_normal(dim).astype(np.float32), agent="kai", session="s1", channel=None,
             importance=0.25, access_count=int(rng.integers(0, 1 << 40)), last_accessed_at="1970-01-01T00:00:00Z",
             created_at="2024-02-29T23:59:59.123456789Z", updated_at="2024-03-01T00:00:00.5+01:30", deleted=False)
    d.update(kw)
    return d


def test_fields_and_unaligned_embedding():
    rng = np.random.default_rng(1)
    for body_len in range(0, 9):  # shifts the embedding through every alignment
        d = _node(rng, body="x" * body_len)
        n = decode_node(B.encode_node(**d))
        assert n["id"] == d["id16"] and n["kind"] == "fact" and n["title"] == d["title"] and n["agent"] == "kai"
        assert np.array_equal(n["embedding"], d["embedding"]) and n["n_tags"] == 2 and not n["deleted"]
        assert n["access_count"] == d["access_count"] and n["importance"] == 0.25
        assert n["created_at"] == (_epoch(2024, 2, 29, 23, 59, 59), 123456789)
        assert n["updated_at"] == (_epoch(2024, 3, 1) - 5400, 500000000)


def test_trailing_bytes_are_allowed_like_bincode_deserialize():
    rng = np.random.default_rng(2)
    rec = B.encode_node(**_node(rng))
    n = decode_node(rec + b"\xff\xff\xff")
    assert n["bytes_used"] == len(rec)


@pytest.mark.parametrize("ts,ok", [
    ("2023-11-14T22:13:20Z", True), ("2023-11-14T22:13:20.000001Z", True), ("2023-11-14t22:13:20z", True),
    ("2023-11-14T22:13:20-08:00", True), ("2023-02-29T00:00:00Z", False), ("2023-11-14T24:00:00Z", False),
    ("2023-11-14T22:13:20", False), ("2023-11-14", False), ("", False), ("2023-11-14T22:13:20.Z", False),
    ("2023-13-01T00:00:00Z", False), ("2016-12-31T23:59:60Z", True),
])
def test_timestamps(ts, ok):
    rng = np.random.default_rng(3)
    rec = B.encode_node(**_node(rng, created_at=ts))
    if ok:
        decode_node(rec)
    else:
        with pytest.raises(ValidationError, match="Failed to deserialize node"):
            decode_node(rec)


def test_records_the_reference_cannot_read_are_errors():
    rng = np.random.default_rng(4)
    d = _node(rng)
    rec = B.encode_node(**d)
    bad = []
    bad.append(rec[:-1])                                   # truncated: no `deleted` byte
    bad.append(rec[:-1] + b"\x02")                         # bool must be 0 | 1
    bad.append(rec[:40])                                   # cut inside the strings
    bad.append(struct.pack("<Q", 15) + rec[8:])            # Uuid length
    i = rec.index(b"fact")
    bad.append(rec[:i] + b"\xff\xfe\xfd\xfc" + rec[i + 4:])  # invalid UTF-8 in kind
    bad.append(rec[:i] + b"\xed\xa0\x80a" + rec[i + 4:])     # surrogate
    bad.append(rec[:i] + b"\xc0\xafab" + rec[i + 4:])        # overlong
    # non-empty data.metadata: serde_json::Value has no bincode decoding (deserialize_any)
    j = 8 + 16 + 8 + 4 + 8 + len(d["title"].encode()) + 8 + len(d["body"])
    assert rec[j:j + 8] == bytes(8)
    bad.append(rec[:j] + struct.pack("<Q", 1) + B._s("k") + B._s("v") + rec[j + 8:])
    # Option tag 2 on the embedding
    e = rec.index(b"\x01" + struct.pack("<Q", 7))
    bad.append(rec[:e] + b"\x02" + rec[e + 1:])
    # embedding length beyond the record
    bad.append(rec[:e + 1] + struct.pack("<Q", 1 << 60) + rec[e + 9:])
    bad.append(b"")
    for b in bad:
        with pytest.raises(ValidationError, match="Failed to deserialize node"):
            decode_node(b)
        with pytest.raises((AssertionError, UnicodeDecodeError, struct.error)):
            B.decode_node(b)  # the Python statement of the layout agrees these are unreadable


def test_decoder_agrees_with_python_statement_on_random_nodes():
    rng = np.random.default_rng(5)
    for _ in range(200):
        d = _node(rng, dim=int(rng.integers(0, 20)), tags=["t%d" % i for i in range(int(rng.integers(0, 4)))],
                  session=None if rng.random() < 0.5 else "sess", channel=None if rng.random() < 0.5 else "slack",
                  deleted=bool(rng.random() < 0.3), embedding=None if rng.random() < 0.3 else rng.standard_normal(5).astype(np.float32))
        rec = B.encode_node(**d)
        a, b = decode_node(rec), B.decode_node(rec)
        assert a["id"] == b["id"] and a["kind"] == b["kind"] and a["agent"] == b["agent"] and a["deleted"] == b["deleted"]
        assert a["n_tags"] == len(b["tags"]) and a["access_count"] == b["access_count"]
        assert (a["embedding"] is None) == (b["embedding"] is None)
        if b["embedding"] is not None:
            assert np.array_equal(a["embedding"], b["embedding"])


def test_decoder_survives_mutations(tmp_path):
    """The decoder reads database bytes, so it is fuzzed: tests/cpp/fuzz_node_decode.cpp builds nodes.cpp's decoder
    alone with AddressSanitizer + UBSan (host only, no device) and feeds it 200k truncated / bit-flipped / length-
    corrupted records held in exact-size heap blocks; any out-of-bounds read or undefined operation aborts."""
    import shutil
    import subprocess
    if shutil.which("g++") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime.h"):
        pytest.skip("needs g++ and the HIP headers")
    root = os.path.dirname(HERE)
    exe = str(tmp_path / "fuzz_node_decode")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{root}/include", f"-I{root}/cortex_amd/csrc",
           f"{root}/tests/cpp/fuzz_node_decode.cpp", "-o", exe, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-lpthread"]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe, "200000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert "mutated records" in r.stdout
