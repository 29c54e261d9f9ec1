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
