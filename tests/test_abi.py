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
