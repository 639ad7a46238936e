"""The Rust adapter (rust/infur-hip-sys) cannot be compiled here (no cargo / rustc in the image), so its FFI surface is tied
to include/infur_hip.h MECHANICALLY: every `pub fn` of the `extern "C"` block must exist in the header with the same arity,
and each parameter / return type must be the Rust spelling of the C type (pointer-ness, constness, integer width,
signedness); both `#[repr(C)]` structs must list the header's fields in order with matching types; the numeric constants
must agree.  This is what tests/test_abi_cpu.py does for the ctypes binding."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "infur_hip.h")).read()
RUST = open(os.path.join(ROOT, "rust", "infur-hip-sys", "src", "lib.rs")).read()

C_SCALARS = {"int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "float": "f32", "double": "f64",
             "uint8_t": "u8", "char": "c_char", "void": "c_void", "int": "i32"}
OPAQUE = {"infur_ctx", "infur_stream", "infur_group", "infur_options", "infur_model_info", "infur_kernel_record"}


def strip_comments(c):
    c = re.sub(r"/\*.*?\*/", " ", c, flags=re.S)
    return re.sub(r"//[^\n]*", " ", c)


def c_type_to_rust(t):
    """'const uint8_t* const*' -> '*const *const u8' (a trailing const on the outermost level is dropped: by-value)"""
    toks = re.findall(r"[A-Za-z_][A-Za-z_0-9]*|\*", t)
    base, consts, levels = None, False, []  # levels: constness of each pointee, innermost first
    cur_const = False
    for tk in toks:
        if tk == "const":
            cur_const = True
        elif tk == "*":
            levels.append(cur_const)
            cur_const = False
        elif tk in ("struct",):
            continue
        else:
            base = tk
    rust = C_SCALARS.get(base, base if base in OPAQUE else None)
    assert rust is not None, f"unknown C type {t!r}"
    for is_const in levels:
        rust = ("*const " if is_const else "*mut ") + rust
    return rust


def header_functions():
    h = strip_comments(HEADER)
    h = h[h.index('extern "C" {'):]
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(infur_\w+)\s*\(([^;{}]*?)\)\s*;", h):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef"):
            continue
        plist = [] if params in ("void", "") else [p.strip() for p in params.split(",")]
        ptypes = []
        for p in plist:
            mm = re.match(r"(.*?)([A-Za-z_]\w*)$", p)  # split off the parameter name
            ptypes.append(c_type_to_rust(mm.group(1)))
        out[name] = (None if ret == "void" else c_type_to_rust(ret), ptypes)
    return out


def rust_functions():
    blk = RUST[RUST.index('extern "C" {'):]
    blk = re.sub(r"//[^\n]*", " ", blk)
    out = {}
    for m in re.finditer(r"pub fn (\w+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+?))?\s*;", blk, flags=re.S):
        name, params, ret = m.group(1), m.group(2), m.group(3)
        ptypes = []
        for p in [x.strip() for x in params.split(",") if x.strip()]:
            ptypes.append(re.sub(r"\s+", " ", p.split(":", 1)[1].strip()))
        out[name] = (re.sub(r"\s+", " ", ret.strip()) if ret else None, ptypes)
    return out


def test_every_rust_extern_matches_the_header():
    hf, rf = header_functions(), rust_functions()
    assert len(rf) >= 38 and len(hf) >= len(rf)
    assert set(hf) == set(rf), f"exports of the header the -sys crate does not bind: {sorted(set(hf) - set(rf))}"
    for name, (rret, rparams) in rf.items():
        assert name in hf, f"{name} is declared in the Rust crate but not in include/infur_hip.h"
        cret, cparams = hf[name]
        assert rret == cret, f"{name}: returns {rret} in Rust, {cret} in C"
        assert len(rparams) == len(cparams), f"{name}: {len(rparams)} parameters in Rust, {len(cparams)} in C"
        for i, (r, c) in enumerate(zip(rparams, cparams)):
            assert r == c, f"{name} parameter {i}: {r} in Rust, {c} in C"


def c_struct_fields(name):
    h = strip_comments(HEADER)
    body = re.search(r"typedef struct %s\s*\{(.*?)\}\s*%s\s*;" % (name, name), h, flags=re.S).group(1)
    out = []
    for decl in [d.strip() for d in body.split(";") if d.strip()]:
        m = re.match(r"(.*?)([A-Za-z_]\w*)((?:\[\d+\])*)$", decl)
        base, fname, dims = c_type_to_rust(m.group(1)), m.group(2), re.findall(r"\[(\d+)\]", m.group(3))
        for d in reversed(dims):  # char a[2][32] -> [[c_char; 32]; 2]
            base = f"[{base}; {d}]"
        out.append((fname, base))
    return out


def rust_struct_fields(name):
    m = re.search(r"#\[repr\(C\)\]\s*pub struct %s\s*\{(.*?)\n\}" % name, RUST, flags=re.S)
    assert m, f"#[repr(C)] struct {name} not found"
    return [(f.group(1), re.sub(r"\s+", " ", f.group(2).strip())) for f in re.finditer(r"pub (\w+)\s*:\s*([^,\n]+),", m.group(1))]


@pytest.mark.parametrize("name", ["infur_options", "infur_model_info", "infur_kernel_record"])
def test_repr_c_structs_match_the_header(name):
    assert rust_struct_fields(name) == c_struct_fields(name)


def test_constants_match_the_header():
    h = strip_comments(HEADER)
    consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(INFUR_[A-Z0-9_]+)\s*=\s*(\d+)", h)}
    consts["INFUR_ABI_VERSION"] = int(re.search(r"#define INFUR_ABI_VERSION (\d+)", HEADER).group(1))
    rs = {m.group(1): int(m.group(2)) for m in re.finditer(r"pub const (INFUR_[A-Z0-9_]+)\s*:\s*[iu]32\s*=\s*(\d+)\s*;", RUST)}
    assert len(rs) >= 19
    for k, v in rs.items():
        assert consts.get(k) == v, (k, v, consts.get(k))
    for k in consts:  # every status / mode of the header is spelled in the crate
        assert k in rs, f"{k} is missing from rust/infur-hip-sys"


def test_adapter_restates_the_provided_generate():
    """processing.rs:53-59: `generate` is a provided method with `Default` bounds on Input and Output"""
    lib = open(os.path.join(ROOT, "rust", "infur-hip", "src", "lib.rs")).read()
    tr = lib[lib.index("pub trait Processor"):]
    tr = tr[:tr.index("\n}\n")]
    assert re.search(r"fn generate\(&mut self\)\s*->\s*Self::ProcessResult\s*where\s*Self::Input:\s*Default,\s*Self::Output:\s*Default,", tr)
    for sig in ("fn control(&mut self, cmd: Self::Command) -> Result<&mut Self, Self::ControlError>;",
                "fn advance(&mut self, inp: &Self::Input, out: &mut Self::Output) -> Self::ProcessResult;", "fn is_dirty(&self) -> bool;"):
        assert sig in tr
