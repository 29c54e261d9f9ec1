"""The Rust side of the boundary (bindings/rust/) against include/dehalo.h -- no rustc exists in the build image, so what can be checked is checked here:

1. bindings/rust/dehalo-sys/src/lib.rs is current: tools/gen_rust_bindings.py regenerates it from the header, byte for byte.
2. With a parser of its OWN (regexes below, sharing no code with the generator): every function the header declares is declared `pub fn` in the .rs file -- same
   name, same number of parameters, and for every parameter and the return value the same width class (pointer / 32-bit int / 64-bit int / usize / f64 / u8) and,
   for pointers, the same const-ness of the pointee; every `#[repr(C)]` struct has the header's fields, in order, with the same classes.
3. The hand-written shim crate (bindings/rust/dehalo-halo2) only calls functions that exist, with the right number of arguments.

SURVEY.md 8(b): "plus a Rust shim (source only)"; reference call sites benches/delay_enc.rs:43-54,84-131; Cargo.toml:10-17."""
import os
import re
import subprocess
import sys

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "dehalo.h")
SYS_RS = os.path.join(ROOT, "bindings", "rust", "dehalo-sys", "src", "lib.rs")
SHIM_DIR = os.path.join(ROOT, "bindings", "rust", "dehalo-halo2", "src")

C_CLASS = {"int": "i32", "uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64", "int64_t": "i64", "size_t": "usize", "double": "f64", "uint8_t": "u8", "char": "i8"}
RS_CLASS = {"c_int": "i32", "u32": "u32", "i32": "i32", "u64": "u64", "i64": "i64", "usize": "usize", "f64": "f64", "u8": "u8", "c_char": "i8"}


def header_text():
    t = open(HEADER).read()
    t = re.sub(r"/\*.*?\*/", "", t, flags=re.S)
    t = "\n".join(l for l in t.splitlines() if not l.lstrip().startswith("#"))
    return t


def c_class(decl, typedef_ints):
    """width class of one C declarator (type + name [+ array]): 'ptr(const)' / 'ptr(mut)' for anything with a * or [], else the integer class; enum typedefs are ints"""
    decl = decl.strip()
    is_ptr = "*" in decl or "[" in decl
    if is_ptr:
        # the OUTERMOST pointer's pointee: `const T* p` / `const T p[4]` -> the const T; `T* const* p` -> the const inner pointer; `const T* const* p` likewise
        segs = decl.split("*")
        stars = len(segs) - 1
        assert not ("[" in decl and stars), decl      # (an array of pointers does not occur in the header)
        pointee = segs[0] if stars <= 1 else segs[stars - 1]
        return "ptr(const)" if re.search(r"\bconst\b", pointee) else "ptr(mut)"
    words = [w for w in re.findall(r"[A-Za-z_][A-Za-z0-9_]*", decl) if w != "const"]
    base = words[0]
    if base in typedef_ints:
        return "i32"
    if base in C_CLASS:
        return C_CLASS[base]
    return "struct:" + base


def c_functions():
    t = header_text()
    enums = set(re.findall(r"typedef enum \{[^}]*\}\s*(\w+)\s*;", t))
    fnptrs = set(re.findall(r"typedef \w+ \(\*(\w+)\)", t))
    out = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(dehalo_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", t):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        if ret.startswith("typedef"):
            continue
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        cls = []
        for p in params:
            c = c_class(p, enums)
            if c.startswith("struct:") and c[7:] in fnptrs:
                c = "fnptr"
            cls.append(c)
        rc = "void" if ret == "void" else c_class(ret + " r", enums)
        out[name] = (rc, cls)
    return out


def rs_class(ty):
    ty = ty.strip()
    if ty.startswith("*const "):
        return "ptr(const)"
    if ty.startswith("*mut "):
        return "ptr(mut)"
    if ty.startswith("Option<"):
        return "fnptr"
    if ty.startswith("["):
        return "array"
    return RS_CLASS.get(ty, "struct:" + ty)


def rs_functions():
    t = open(SYS_RS).read()
    fn_aliases = set(re.findall(r"pub type (\w+) = Option<unsafe extern", t))      # a callback passed as a parameter is named by its alias
    out = {}
    for m in re.finditer(r"pub fn (dehalo_[a-z0-9_]+)\(([^)]*)\)(?:\s*->\s*([^;]+))?;", t):
        name, args, ret = m.group(1), m.group(2).strip(), m.group(3)
        params = [] if not args else [a.split(":", 1)[1] for a in args.split(", ")]
        out[name] = ("void" if ret is None else rs_class(ret), ["fnptr" if p.strip() in fn_aliases else rs_class(p) for p in params])
    return out


def test_generated_file_is_current():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_bindings.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_every_header_function_is_declared_with_the_same_shape():
    c, rs = c_functions(), rs_functions()
    assert len(c) >= 100, len(c)
    assert sorted(c) == sorted(rs), (sorted(set(c) - set(rs)), sorted(set(rs) - set(c)))
    n_pub = len(re.findall(r"pub fn dehalo_", open(SYS_RS).read()))
    assert n_pub == len(c)                                      # `grep -c 'pub fn dehalo_' src/lib.rs` = the header's count
    assert ("pub const DEHALO_SYS_FUNCTIONS: usize = %d;" % len(c)) in open(SYS_RS).read()
    for name, (rc, params) in c.items():
        rrc, rparams = rs[name]
        assert len(params) == len(rparams), (name, "arity", params, rparams)
        assert rc == rrc, (name, "return", rc, rrc)
        for i, (a, b) in enumerate(zip(params, rparams)):
            assert a == b, (name, "parameter %d" % i, a, b)
    # the library exports them all (the binding's own check lives in tests/test_abi.py; here: the same set of names)
    from test_abi import _declared_symbols
    assert sorted(c) == _declared_symbols()


def test_repr_c_structs_have_the_headers_fields_in_order():
    t = header_text()
    enums = set(re.findall(r"typedef enum \{[^}]*\}\s*(\w+)\s*;", t))
    rs = open(SYS_RS).read()
    n = 0
    for m in re.finditer(r"typedef struct \{(.*?)\}\s*(\w+)\s*;", t, flags=re.S):
        body, name = m.group(1), m.group(2)
        fields = []
        for stmt in [s.strip() for s in body.split(";") if s.strip()]:
            parts = [p.strip() for p in stmt.split(",")]
            first = parts[0]
            base_is_const = bool(re.match(r"^const\b", first))
            base = [w for w in re.findall(r"[A-Za-z_][A-Za-z0-9_]*", first) if w != "const"][0]
            for i, p in enumerate(parts):
                fname = re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[\d+\])?$", p)[0]
                if "[" in p and "*" not in p:
                    cls = "array"
                elif "*" in p:
                    cls = "ptr(const)" if base_is_const else "ptr(mut)"
                elif base in enums:
                    cls = "i32"
                else:
                    cls = C_CLASS.get(base, "struct:" + base)
                fields.append((fname, cls))
        rm = re.search(r"#\[repr\(C\)\]\n(?:#\[derive\([^\n]*\)\]\n)?pub struct %s \{(.*?)\n\}" % name, rs, flags=re.S)
        assert rm, "no #[repr(C)] struct " + name
        rfields = [(a, rs_class(b)) for a, b in re.findall(r"pub (\w+): ([^\n]+),", rm.group(1))]
        rename = {"in": "input"}
        want = [(rename.get(a, a), b) for a, b in fields]
        assert want == rfields, (name, want, rfields)
        n += 1
    assert n >= 12
    for opaque in re.findall(r"typedef struct (\w+) \1;", t):
        assert re.search(r"#\[repr\(C\)\]\npub struct %s \{\n    _private: \[u8; 0\],\n\}" % opaque, rs), opaque


def test_shim_crate_calls_only_what_exists_with_the_right_arity():
    """bindings/rust/dehalo-halo2: every `sys::dehalo_*(...)` call names a declared function and passes as many arguments as it takes; the functions the verdict
    names are there: best_multiexp, best_fft, ParamsKZG::{setup, commit, commit_lagrange}, create_proof."""
    decl = rs_functions()
    calls = 0
    for fn in sorted(os.listdir(SHIM_DIR)):
        src = open(os.path.join(SHIM_DIR, fn)).read()
        src = re.sub(r"//[^\n]*", "", src)
        for m in re.finditer(r"sys::(dehalo_[a-z0-9_]+)\s*\(", src):
            name = m.group(1)
            assert name in decl, (fn, name)
            depth, i, args, cur = 1, m.end(), 0, ""
            while depth:      # balanced-parenthesis scan, top-level commas
                ch = src[i]
                if ch in "([{":
                    depth += 1
                elif ch in ")]}":
                    depth -= 1
                elif ch == "," and depth == 1:
                    args += 1
                    cur = ""
                    i += 1
                    continue
                if depth:
                    cur += ch
                i += 1
            n_args = args + (1 if cur.strip() else 0)
            assert n_args == len(decl[name][1]), (fn, name, n_args, len(decl[name][1]))
            calls += 1
    assert calls >= 25
    text = {fn: open(os.path.join(SHIM_DIR, fn)).read() for fn in os.listdir(SHIM_DIR)}
    assert "pub fn best_multiexp<" in text["arithmetic.rs"] and "pub fn best_fft<" in text["arithmetic.rs"]
    for f in ("pub fn setup<", "pub fn commit(", "pub fn commit_lagrange(", "pub fn read(", "pub fn write("):
        assert f in text["params.rs"], f
    assert "pub fn create_proof<" in text["prover.rs"] and "pub fn keygen(" in text["prover.rs"]
    cargo = open(os.path.join(ROOT, "bindings", "rust", "dehalo-halo2", "Cargo.toml")).read()
    assert 'tag = "v2023_04_20"' in cargo and "dehalo-sys" in cargo          # the reference's pin (Cargo.toml:17)
