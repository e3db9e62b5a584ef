"""CPU: the C-ABI library loads, exports every function include/bpg.h declares,
the Python binding covers them all, and the product fails loudly without a GPU
(no CPU fallback).  No compute calls."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "bpg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bpg_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    lib = ctypes.CDLL(os.path.join(ROOT, "bulletproofs_gadgets_amd", "libbpg.so"))
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "libbpg.so does not export %s" % n


def test_binding_covers_the_header():
    from bulletproofs_gadgets_amd import _native
    assert set(_declared()) == set(_native.exported_symbols())


def test_header_cites_reference_call_sites():
    text = open(os.path.join(ROOT, "include", "bpg.h")).read()
    assert "tests/edwards_point.rs" in text and "Replaces" in text


def test_no_gpu_means_loud_failure():
    import bulletproofs_gadgets_amd as bpg
    lib = bpg._native.lib()
    n = ctypes.c_int(0)
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        have_gpu = hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        have_gpu = False
    if have_gpu:
        pytest.skip("a GPU is present")
    with pytest.raises(bpg.BpgError) as e:
        bpg.Context(0)
    assert e.value.code == bpg._native.EDEVICE
    assert b"no CPU fallback" in lib.bpg_last_error()


def _c_prototypes():
    """{name: number of parameters} for every function include/bpg.h declares"""
    text = open(os.path.join(ROOT, "include", "bpg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"typedef[^;]*\(\*[^;]*;", "", text)                     # function-pointer typedefs are not entry points
    out = {}
    for name, args in re.findall(r"\b(bpg_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = args.strip()
        out[name] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def test_rust_sys_crate_declares_the_header():
    """rust/bpg-sys/src/lib.rs declares every entry point of include/bpg.h with the same number of
    parameters (no Rust toolchain in this image: names and arities are what can be checked here)."""
    src = open(os.path.join(ROOT, "rust", "bpg-sys", "src", "lib.rs")).read()
    src = re.sub(r"//[^\n]*", "", src)
    rust = {}
    for name, args in re.findall(r"pub fn (bpg_[a-z0-9_]+)\s*\(([^)]*)\)", src, flags=re.S):
        args = args.strip().rstrip(",")
        rust[name] = 0 if not args else len(re.findall(r"[A-Za-z_][A-Za-z0-9_]*\s*:", args))
    c = _c_prototypes()
    assert set(c) == set(_declared())
    assert set(rust) == set(c), (sorted(set(c) - set(rust)), sorted(set(rust) - set(c)))
    for name, n in c.items():
        assert rust[name] == n, "%s: %d parameters in bpg.h, %d in bpg-sys" % (name, n, rust[name])
    # the safe wrapper uses only what the -sys crate declares
    used = set()
    for f in ("lib.rs", "r1cs.rs", "gadgets.rs"):
        used |= set(re.findall(r"sys::(bpg_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "rust", "bpg", "src", f)).read()))
    assert used and used <= set(rust), sorted(used - set(rust))
