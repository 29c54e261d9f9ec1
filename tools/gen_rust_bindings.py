#!/usr/bin/env python3
"""include/dehalo.h -> bindings/rust/dehalo-sys/src/lib.rs: every exported function as an `extern "C"` declaration, every struct as `#[repr(C)]`,
every enumerator as a constant.  The header is the single source; the .rs file is generated, committed, and checked by tests/test_rust_binding.py
(regenerated text == committed text, and -- with a parser of its own -- name, arity and the width of every parameter against the header).

    python tools/gen_rust_bindings.py            # rewrites bindings/rust/dehalo-sys/src/lib.rs
    python tools/gen_rust_bindings.py --check    # exit 1 if the committed file is stale

No rustc exists in the build image: the output is source for the maintainer of the reference's Cargo.toml:17 dependency."""
from __future__ import annotations

import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dehalo.h")
OUT = os.path.join(ROOT, "bindings", "rust", "dehalo-sys", "src", "lib.rs")

BASE = {"int": "c_int", "char": "c_char", "void": "c_void", "size_t": "usize", "double": "f64", "uint64_t": "u64", "uint32_t": "u32", "int32_t": "i32",
        "uint8_t": "u8", "int64_t": "i64"}
RUST_KEYWORDS = {"in": "input", "type": "ty", "fn": "func", "ref": "reference", "mod": "modulus", "box": "boxed", "move": "mv", "match": "matched", "use": "used", "as": "as_"}


def strip(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = "\n".join(l for l in text.splitlines() if not l.strip().startswith("#"))
    text = text.replace('extern "C" {', "")
    return text


def statements(text: str):
    """top-level statements, split at ';' outside braces"""
    depth, cur = 0, []
    for ch in text:
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
            if depth < 0:      # the closing brace of extern "C"
                depth = 0
                continue
        if ch == ";" and depth == 0:
            s = " ".join("".join(cur).split())
            if s:
                yield s
            cur = []
        else:
            cur.append(ch)


def rust_type(spec: str, stars: list, array: bool, const_base: bool) -> str:
    """spec: the base type name; stars: per '*' whether a `const` FOLLOWS it; array: a trailing [N] on a parameter (decays to a pointer)"""
    t = BASE.get(spec, spec)
    pointee_const = const_base
    for follows_const in stars:
        t = ("*const " if pointee_const else "*mut ") + t
        pointee_const = follows_const
    if array:
        t = ("*const " if pointee_const else "*mut ") + t
    return t


def parse_decl(decl: str):
    """`const uint64_t* const* name` / `uint64_t out[12]` / `int kind` -> (rust type, name, array_len or None)"""
    decl = decl.strip()
    m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)\s*(\[\s*(\d*)\s*\])?$", decl)
    assert m, decl
    head, name, arr, arr_n = m.group(1).strip(), m.group(2), m.group(3), m.group(4)
    toks = re.findall(r"[A-Za-z_][A-Za-z0-9_]*|\*", head)
    const_base, spec, stars = False, None, []
    for t in toks:
        if t == "const":
            if stars:
                stars[-1] = True
            else:
                const_base = True
        elif t == "*":
            stars.append(False)
        elif t in ("struct", "unsigned"):
            continue
        else:
            assert spec is None, decl
            spec = t
    assert spec, decl
    return spec, stars, const_base, name, (int(arr_n) if arr and arr_n else (0 if arr else None))


def param(decl: str):
    spec, stars, const_base, name, arr = parse_decl(decl)
    return RUST_KEYWORDS.get(name, name), rust_type(spec, stars, arr is not None, const_base)


def split_args(args: str):
    args = args.strip()
    if args in ("", "void"):
        return []
    return [a.strip() for a in args.split(",")]


def field_decls(body: str):
    """struct body -> [(name, rust type)]; `const uint64_t *beta, *gamma;` and `uint64_t a[2], b[2];` declare several"""
    out = []
    for stmt in [s.strip() for s in body.split(";") if s.strip()]:
        parts = [p.strip() for p in stmt.split(",")]
        spec0, stars0, const0, name0, arr0 = parse_decl(parts[0])
        decls = [(stars0, name0, arr0)]
        for p in parts[1:]:
            m = re.match(r"^((?:\*\s*(?:const\s*)?)*)([A-Za-z_][A-Za-z0-9_]*)\s*(\[\s*(\d+)\s*\])?$", p)
            assert m, (stmt, p)
            st = [bool(re.match(r"\*\s*const", x)) for x in re.findall(r"\*\s*(?:const)?", m.group(1))]
            decls.append((st, m.group(2), int(m.group(4)) if m.group(3) else None))
        for st, name, arr in decls:
            t = rust_type(spec0, st, False, const0)
            if arr is not None:
                t = "[%s; %d]" % (t, arr)
            out.append((RUST_KEYWORDS.get(name, name), t))
    return out


def enumerators(body: str):
    vals, nxt = [], 0
    for item in [x.strip() for x in body.split(",") if x.strip()]:
        if "=" in item:
            n, v = [x.strip() for x in item.split("=")]
            nxt = int(v, 0)
        else:
            n = item
        vals.append((n, nxt))
        nxt += 1
    return vals


def generate() -> str:
    text = strip(open(HEADER).read())
    consts, opaque, structs, fnptrs, funcs = [], [], [], [], []
    for s in statements(text):
        m = re.match(r"^typedef enum \{(.*)\} (\w+)$", s)
        if m:
            consts.append((m.group(2), "c_int", enumerators(m.group(1))))
            continue
        m = re.match(r"^enum \{(.*)\}$", s)
        if m:
            consts.append((None, "u32", enumerators(m.group(1))))
            continue
        m = re.match(r"^typedef struct (\w+) (\w+)$", s)
        if m:
            assert m.group(1) == m.group(2)
            opaque.append(m.group(1))
            continue
        m = re.match(r"^typedef struct \{(.*)\} (\w+)$", s)
        if m:
            structs.append((m.group(2), field_decls(m.group(1))))
            continue
        m = re.match(r"^typedef (\w+) \(\*(\w+)\)\((.*)\)$", s)
        if m:
            fnptrs.append((m.group(2), BASE[m.group(1)], [param(a) for a in split_args(m.group(3))]))
            continue
        m = re.match(r"^(.*?)\b(dehalo_\w+)\s*\((.*)\)$", s)
        assert m, "unparsed statement: " + s
        ret = m.group(1).strip()
        rspec, rstars, rconst, _, _ = parse_decl(ret + " r") if ret != "void" else ("void", [], False, None, None)
        rt = None if ret == "void" else rust_type(rspec, rstars, False, rconst)
        funcs.append((m.group(2), rt, [param(a) for a in split_args(m.group(3))]))

    o = []
    o.append("//! dehalo-sys: raw FFI declarations of libdehalo.so (include/dehalo.h), the MI355X back end behind halo2_proofs' `create_proof` / `commit` /")
    o.append("//! `best_multiexp` / `best_fft` (reference call sites: benches/delay_enc.rs:43-54,84-131 of radiusxyz/delay-encryption-in-halo2).")
    o.append("//!")
    o.append("//! GENERATED by tools/gen_rust_bindings.py from include/dehalo.h -- do not edit; tests/test_rust_binding.py fails when this file is stale or when a")
    o.append("//! declaration disagrees with the header in name, arity or the width of a parameter.  The semantics of every entry point are documented in the header.")
    o.append("#![allow(non_camel_case_types, non_upper_case_globals, clippy::too_many_arguments)]")
    o.append("#![no_std]")
    o.append("")
    o.append("pub use core::ffi::{c_char, c_int, c_void};")
    o.append("")
    o.append("// ---- enumerators -------------------------------------------------------------------------------------------------")
    for tname, ty, vals in consts:
        if tname:
            o.append("pub type %s = c_int;" % tname)
        for n, v in vals:
            o.append("pub const %s: %s = %d;" % (n, tname or ty, v))
        o.append("")
    o.append("// ---- opaque handles ----------------------------------------------------------------------------------------------")
    for n in opaque:
        o.append("#[repr(C)]\npub struct %s {\n    _private: [u8; 0],\n}" % n)
    o.append("")
    o.append("// ---- callbacks ---------------------------------------------------------------------------------------------------")
    for n, rt, ps in fnptrs:
        o.append("pub type %s = Option<unsafe extern \"C\" fn(%s) -> %s>;" % (n, ", ".join("%s: %s" % p for p in ps), rt))
    o.append("")
    o.append("// ---- structs (field for field as in the header) --------------------------------------------------------------------")
    for n, fields in structs:
        o.append("#[repr(C)]\n#[derive(Clone, Copy)]\npub struct %s {" % n)
        for fn_, ft in fields:
            o.append("    pub %s: %s," % (fn_, ft))
        o.append("}")
    o.append("")
    o.append("// ---- functions (%d) -------------------------------------------------------------------------------------------------" % len(funcs))
    o.append("#[link(name = \"dehalo\")]")
    o.append("extern \"C\" {")
    for n, rt, ps in funcs:
        sig = "    pub fn %s(%s)%s;" % (n, ", ".join("%s: %s" % p for p in ps), " -> " + rt if rt else "")
        o.append(sig)
    o.append("}")
    o.append("")
    o.append("/// number of functions declared above (= the header's count; tests/test_rust_binding.py)")
    o.append("pub const DEHALO_SYS_FUNCTIONS: usize = %d;" % len(funcs))
    return "\n".join(o) + "\n"


if __name__ == "__main__":
    text = generate()
    if "--check" in sys.argv:
        ok = os.path.exists(OUT) and open(OUT).read() == text
        print("bindings/rust/dehalo-sys/src/lib.rs is %s" % ("current" if ok else "STALE: run tools/gen_rust_bindings.py"))
        sys.exit(0 if ok else 1)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write(text)
    print("wrote %s (%d lines)" % (os.path.relpath(OUT, ROOT), text.count("\n")))
