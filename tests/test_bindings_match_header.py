"""Mechanical guard against drift between include/cortex_hip.h and the two hand-written bindings of it:
the Rust extern block a maintainer drops into cortex-core (integration/hip_index.rs — not compilable in this image,
so nothing else would notice a stale signature) and the ctypes table the parity tests call through
(cortex_amd/_lib.py).  Every prototype is parsed from the header; each binding must name the same functions with the
same number of arguments, the same width / pointer-ness / constness per argument, and the same return type."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strip_c(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return re.sub(r"^\s*#.*$", "", text, flags=re.M)


def _canon_c(t):
    """C type -> (kind, const?) with kind in {i32,u32,i64,u64,f32,f64,ptr,void}"""
    t = re.sub(r"\s+", " ", t.replace("*", " * ")).strip()
    if "*" in t or t.endswith("]"):
        base = t.split("*")[0]
        return ("ptr", "const" in base.split() or (t.endswith("]") and "const" in t.split()))
    t = t.replace("const ", "").strip()
    return ({"int": "i32", "int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "float": "f32",
             "double": "f64", "void": "void", "size_t": "u64"}[t], False)


def c_prototypes():
    protos = {}
    for hdr in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if not hdr.endswith(".h"):
            continue
        text = _strip_c(open(os.path.join(ROOT, "include", hdr)).read())
        text = re.sub(r"typedef struct \w+ \{.*?\} \w+;", "", text, flags=re.S)
        for m in re.finditer(r"([A-Za-z_][\w \*]*?)\b(cx_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
            ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
            params = []
            if args and args != "void":
                for a in args.split(","):
                    a = a.strip()
                    arr = a.endswith("]")
                    a2 = re.sub(r"\[\d*\]$", "", a)
                    ty = re.sub(r"\b[A-Za-z_]\w*$", "", a2).strip() if not re.fullmatch(r"[\w ]*\*", a2) else a2
                    params.append(_canon_c(ty + (" []" if arr else "")))
            protos[name] = (_canon_c(ret) if ret != "void" else ("void", False), params)
    return protos


def _canon_rust(t):
    t = t.strip()
    if t.startswith("*const"):
        return ("ptr", True)
    if t.startswith("*mut"):
        return ("ptr", False)
    return ({"c_int": "i32", "i32": "i32", "u32": "u32", "i64": "i64", "u64": "u64", "f32": "f32", "f64": "f64",
             "usize": "u64"}[t], False)


def rust_externs():
    text = open(os.path.join(ROOT, "integration", "hip_index.rs")).read()
    text = re.sub(r"//[^\n]*", "", text)
    block = re.search(r'extern "C" \{(.*?)\n\}', text, flags=re.S).group(1)
    out = {}
    for m in re.finditer(r"fn (cx_\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        params = [_canon_rust(a.split(":", 1)[1]) for a in m.group(2).split(",") if a.strip()]
        out[m.group(1)] = (_canon_rust(m.group(3)) if m.group(3) else ("void", False), params)
    return out


def test_header_parser_sees_the_abi():
    p = c_prototypes()
    assert len(p) >= 45
    assert p["cx_create"] == (("ptr", False), [("u32", False), ("i32", False)])
    assert p["cx_last_error"] == (("ptr", True), [])
    assert p["cx_upsert"][1] == [("ptr", False), ("ptr", True), ("ptr", True), ("u64", False)]
    assert p["cx_search"][1][0] == ("ptr", True)       # &self
    assert p["cx_rebuild"][1][0] == ("ptr", False)     # &mut self


def test_rust_extern_block_matches_the_header():
    c, r = c_prototypes(), rust_externs()
    assert len(r) >= 25
    for name, (rret, rparams) in r.items():
        assert name in c, f"hip_index.rs binds {name}, which include/cortex_hip.h does not declare"
        cret, cparams = c[name]
        assert len(rparams) == len(cparams), f"{name}: {len(rparams)} arguments in Rust, {len(cparams)} in the header"
        for i, (rp, cp) in enumerate(zip(rparams, cparams)):
            assert rp[0] == cp[0], f"{name} argument {i}: Rust {rp[0]}, header {cp[0]}"
            if rp[0] == "ptr":
                assert rp[1] == cp[1], f"{name} argument {i}: constness differs (Rust const={rp[1]}, header const={cp[1]})"
        assert rret[0] == cret[0], f"{name}: return type Rust {rret}, header {cret}"
    # the trait's methods and the linker passes must all be bound (vector/index.rs:50-99; auto_linker.rs:215-264; dedup.rs:65-127)
    for need in ("cx_create", "cx_destroy", "cx_upsert", "cx_remove", "cx_search", "cx_search_threshold", "cx_search_batch",
                 "cx_len", "cx_rebuild", "cx_save", "cx_load", "cx_set_metadata", "cx_lookup", "cx_autolink_pass_rows",
                 "cx_dedup_scan_rows", "cx_topk_lists_rows", "cx_last_error"):
        assert need in r, f"hip_index.rs does not bind {need}"


def _struct_fields_c(name):
    text = _strip_c(open(os.path.join(ROOT, "include", "cortex_hip.h")).read())
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, flags=re.S).group(1)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = [x.strip() for x in decl.split(",")]          # `double a, b, c` declares three fields of one type
        first = names[0]
        arr = re.search(r"\[(\d+)\]$", first)
        d2 = re.sub(r"\[\d+\]$", "", first)
        ty = re.sub(r"\b[A-Za-z_]\w*$", "", d2).strip()
        kind = ("bytes%s" % arr.group(1)) if arr else _canon_c(ty)[0]
        out += [kind] * len(names)
    return out


def _struct_fields_rust(name):
    text = open(os.path.join(ROOT, "integration", "hip_index.rs")).read()
    text = re.sub(r"//[^\n]*", "", text)
    m = re.search(r"#\[repr\(C\)\][^{]*struct %s \{(.*?)\}" % name, text, flags=re.S)
    out = []
    for f in m.group(1).split(","):
        if ":" in f:
            out.append(_canon_rust(f.split(":", 1)[1])[0])
    return out


def test_rust_repr_c_structs_match_the_header():
    for c_name, r_name in (("cx_filter", "CxFilter"), ("cx_decay_config", "CxDecayConfig"), ("cx_bulk_stats", "CxBulkStats")):
        assert _struct_fields_rust(r_name) == _struct_fields_c(c_name), f"{r_name} vs {c_name}"


def test_ctypes_table_matches_the_header():
    import ctypes as C
    from cortex_amd import _lib
    c = c_prototypes()
    width = {"i32": 4, "u32": 4, "i64": 8, "u64": 8, "f32": 4, "f64": 8}
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        cret, cparams = c[name]
        assert len(argtypes) == len(cparams), f"{name}: ctypes table has {len(argtypes)} arguments, header {len(cparams)}"
        for i, (a, cp) in enumerate(zip(argtypes, cparams)):
            is_ptr = a in (C.c_void_p, C.c_char_p) or hasattr(a, "contents") or getattr(a, "_type_", None) == "P"
            if cp[0] == "ptr":
                assert is_ptr, f"{name} argument {i}: header has a pointer, ctypes {a}"
            else:
                assert not is_ptr and C.sizeof(a) == width[cp[0]], f"{name} argument {i}: header {cp[0]}, ctypes {a}"
                assert (a in (C.c_float, C.c_double)) == (cp[0] in ("f32", "f64")), f"{name} argument {i}: float/int mismatch"
        if cret[0] == "void":
            assert restype is None
        elif cret[0] == "ptr":
            assert restype in (C.c_void_p, C.c_char_p)
        else:
            assert C.sizeof(restype) == width[cret[0]] and (restype in (C.c_float, C.c_double)) == (cret[0] in ("f32", "f64"))
