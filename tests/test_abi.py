"""CPU suite: the C-ABI library loads, exports every symbol include/dehalo.h declares, and
refuses to run without a device (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "dehalo.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dehalo_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported(pkg):
    lib = pkg.load_library()
    syms = _declared_symbols()
    assert len(syms) >= 24
    for s in syms:
        assert hasattr(lib, s), "libdehalo.so does not export " + s
    # and the Python binding knows every one of them
    from importlib import import_module
    import sys
    binding = sys.modules["dehalo2_amd._lib"]
    assert sorted(binding.SYMBOLS) == syms


def test_every_binding_declares_its_argument_types(pkg):
    """A ctypes call without argtypes passes a Python int as a 32-bit C int: a device pointer would be truncated and the kernel
    would fault.  Every bound entry point that takes arguments must declare them."""
    import sys
    lib = pkg.load_library()
    binding = sys.modules["dehalo2_amd._lib"]
    for name in binding.SYMBOLS:
        if name == "dehalo_version":
            continue
        assert getattr(lib, name).argtypes, name + " has no argtypes"


def test_version_string(pkg):
    assert pkg.load_library().dehalo_version().decode().startswith("dehalo")


def test_no_device_means_error_not_fallback(pkg):
    """On a box without a GPU ctx_create must fail with DEHALO_ERR_NO_DEVICE; on a GPU box an
    out-of-range ordinal must."""
    lib = pkg.load_library()
    h = C.c_void_p()
    rc = lib.dehalo_ctx_create(1 << 20, C.byref(h))
    assert rc == -2 and not h.value
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if not has_gpu:
        with pytest.raises(pkg.DehaloError):
            pkg.Context(0)


def test_null_context_is_rejected(pkg):
    lib = pkg.load_library()
    assert lib.dehalo_ctx_synchronize(None) == -1
    assert lib.dehalo_timing_enable(None, 1) == -1
    assert lib.dehalo_last_error(None) == b"null context"


def test_product_does_not_touch_the_oracle():
    """The product path must never import, link or execute anything under oracle/."""
    pkg_dir = os.path.join(ROOT, "delay-encryption-in-halo2_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cuh", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in txt.lower(), (fn, "mentions the oracle")


def test_default_build_reads_no_tuning_from_the_environment(pkg):
    """An environment variable must not change which kernels a drop-in library runs: the shipped library holds the names of three host-side diagnostics
    (include/dehalo.h) and of no A/B switch -- those exist only under `make EXPERIMENTS=1` -- and the sources read the environment nowhere else."""
    blob = open(pkg.library_path(), "rb").read()
    names = sorted(set(m.decode() for m in re.findall(rb"DEHALO_[A-Z0-9_]{3,}", blob)))
    env_like = [n for n in names if not n.startswith(("DEHALO_ERR_", "DEHALO_K_", "DEHALO_OK"))]
    assert set(env_like) <= {"DEHALO_PROVER_TRACE", "DEHALO_SYNTH_THREADS", "DEHALO_SYNTH_TRACE"}, env_like
    allowed = {"DEHALO_PROVER_TRACE", "DEHALO_SYNTH_THREADS", "DEHALO_SYNTH_TRACE"}
    csrc = os.path.join(ROOT, "delay-encryption-in-halo2_amd", "csrc")
    for fn in sorted(os.listdir(csrc)):
        if not fn.endswith((".hip", ".cuh", ".hpp", ".h")):
            continue
        depth = 0      # inside #ifdef DEHALO_EXPERIMENTS / #if DEHALO_PHASE_STAMPS blocks a plain getenv is the measurement build's
        for line in open(os.path.join(csrc, fn)):
            t = line.strip()
            if t.startswith("#if"):
                depth = depth + 1 if (depth or "DEHALO_EXPERIMENTS" in t or "DEHALO_PHASE_STAMPS" in t) and not t.startswith("#ifndef") else depth
            elif t.startswith("#endif") and depth:
                depth -= 1
            for name in re.findall(r'(?<![A-Z_])getenv\("([A-Z0-9_]+)"\)', line):
                assert depth or name in allowed, "%s reads %s from the environment in the default build" % (fn, name)


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: include/dehalo.h compiles as C99 (pedantic, warnings are errors) and as C++11, and a C program that only includes it and takes the
    address of every declared function links against the library -- what a cgo / JNI / Rust `extern "C"` binding relies on."""
    import subprocess
    hdr = os.path.join(ROOT, "include", "dehalo.h")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-x", "c", hdr])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", hdr])
    src = tmp_path / "link_all.c"
    syms = _declared_symbols()
    src.write_text('#include "dehalo.h"\n#include <stdio.h>\nint main(void) {\n    const void* f[] = {%s};\n    printf("%%u\\n", (unsigned)(sizeof f / sizeof f[0]));\n    return f[0] == 0;\n}\n' %
                   ", ".join("(const void*)%s" % s for s in syms))
    exe = tmp_path / "link_all"
    libdir = os.path.join(ROOT, "delay-encryption-in-halo2_amd")
    subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-Werror", "-Wno-pedantic", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src), "-L", libdir, "-ldehalo",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and int(out.stdout.strip()) == len(syms), (out.returncode, out.stdout, out.stderr)
