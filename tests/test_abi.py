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
    for f in ("lib.rs", "r1cs.rs", "fast.rs"):
        used |= set(re.findall(r"sys::(bpg_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "rust", "bulletproofs", "src", f)).read()))
    assert used and used <= set(rust), sorted(used - set(rust))


# ---- crates named `bulletproofs` and `merlin` that the reference's own files build against unchanged.
# No Rust toolchain here: names, trait methods and arities are checked by regex against the reference's call sites.
# What the reference imports and calls (ref file:line for each):
_REF_IMPORTS = {
    # use bulletproofs::r1cs::{...}: src/gadgets/boolean.rs:1-4, scalar.rs:1-3, point/edwards_point.rs:2-4,
    # point/ristretto_point.rs:2-4, sk_knowledge/sk_know_gadget.rs:5-8, tests/edwards_point.rs:8
    "r1cs.rs": ["ConstraintSystem", "LinearCombination", "Variable", "Prover", "Verifier", "R1CSProof", "R1CSError",
                "RandomizedConstraintSystem"],
    # use bulletproofs::{BulletproofGens, PedersenGens}: tests/edwards_point.rs:9, scalar.rs:4, boolean.rs:5
    "lib.rs": ["BulletproofGens", "PedersenGens"],
}
# method name -> number of parameters besides self, as the reference calls it
_REF_CALLS = {
    "multiply": 2,               # cs.multiply(lc, lc)                          edwards_point.rs:45
    "constrain": 1,              # cs.constrain(lc)                             boolean.rs:20
    "allocate_multiplier": 1,    # cs.allocate_multiplier(Option<(Scalar, Scalar)>)   scalar.rs:10-21
    "commit": None,              # prover.commit(v, blinding) / verifier.commit(V): checked per struct below
    "prove": 1,                  # prover.prove(&bp_gens)                       scalar.rs:46
    "verify": 4,                 # verifier.verify(&proof, &pc_gens, &bp_gens, &mut rng)   tests/edwards_point.rs:79
}


def _rust(path):
    return re.sub(r"//[^\n]*", "", open(os.path.join(ROOT, "rust", *path)).read())


def _fn_arity(src, name):
    """parameters besides self of every `fn name(` in src"""
    out = []
    for m in re.finditer(r"fn %s\s*(?:<[^>]*>)?\s*\(" % name, src):
        depth, args, cur = 1, [], ""
        for ch in src[m.end():]:                       # split the parameter list at top-level commas only
            if ch in "([<":
                depth += 1
            elif ch in ")]>":
                depth -= 1
                if depth == 0:
                    break
            if ch == "," and depth == 1:
                args.append(cur)
                cur = ""
            else:
                cur += ch
        args.append(cur)
        args = [a for a in args if a.strip()]
        out.append(len([a for a in args if not re.match(r"\s*(&\s*(mut\s+)?)?(mut\s+)?self\b", a)]))
    return out


def test_standin_crates_are_named_and_shaped_like_the_references_dependencies():
    cargo = open(os.path.join(ROOT, "rust", "bulletproofs", "Cargo.toml")).read()
    assert re.search(r'^name\s*=\s*"bulletproofs"', cargo, flags=re.M) and "yoloproofs" in cargo
    assert re.search(r'^name\s*=\s*"merlin"', open(os.path.join(ROOT, "rust", "merlin", "Cargo.toml")).read(), flags=re.M)
    for f, names in _REF_IMPORTS.items():
        src = _rust(("bulletproofs", "src", f))
        for n in names:
            assert re.search(r"pub (struct|enum|trait) %s\b" % n, src), "%s: no public %s" % (f, n)
    assert "pub mod r1cs;" in _rust(("bulletproofs", "src", "lib.rs"))
    r1cs = _rust(("bulletproofs", "src", "r1cs.rs"))
    for name, want in _REF_CALLS.items():
        got = _fn_arity(r1cs, name)
        assert got, "r1cs.rs has no fn %s" % name
        if want is not None:
            assert want in got, (name, got)
    assert sorted(_fn_arity(r1cs, "commit")) == [1, 2]                 # Verifier::commit(V), Prover::commit(v, v_blinding)
    assert 2 in _fn_arity(r1cs, "new") and 1 in _fn_arity(r1cs, "new")  # Prover::new(pc_gens, transcript), Verifier::new(transcript)
    # R1CSError::VerificationError is matched by name (ref: tests/edwards_point.rs:346)
    assert re.search(r"enum R1CSError\s*\{[^}]*VerificationError", r1cs, flags=re.S)
    # LinearCombination: From<Scalar>, From<Variable>, + and - with anything that converts (ref: boolean.rs:13-16, edwards_point.rs:55-79)
    for pat in (r"impl From<Variable> for LinearCombination", r"impl From<Scalar> for LinearCombination",
                r"impl<L: Into<LinearCombination>> Add<L> for LinearCombination", r"impl<L: Into<LinearCombination>> Sub<L> for LinearCombination",
                r"impl<L: Into<LinearCombination>> Add<L> for Variable", r"impl<L: Into<LinearCombination>> Sub<L> for Variable"):
        assert re.search(pat, r1cs), pat
    # merlin::Transcript::new(label) (ref: tests/edwards_point.rs:25)
    merlin = _rust(("merlin", "src", "lib.rs"))
    assert re.search(r"pub struct Transcript\b", merlin) and 1 in _fn_arity(merlin, "new")
    # BulletproofGens::new(gens_capacity, party_capacity), PedersenGens::default() (ref: tests/edwards_point.rs:89-90)
    lib = _rust(("bulletproofs", "src", "lib.rs"))
    assert 2 in _fn_arity(lib, "new") and re.search(r"impl Default for PedersenGens", lib)


def test_reference_call_sites_use_only_what_the_standins_offer():
    """every `cs.<method>(`, `prover.<method>(`, `verifier.<method>(` in the reference's gadget sources and tests names a
    method the stand-in r1cs module defines (skipped where /root/reference is absent: the GPU box)"""
    ref = "/root/reference"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present")
    r1cs = _rust(("bulletproofs", "src", "r1cs.rs"))
    defined = set(re.findall(r"fn ([a-z_0-9]+)\s*(?:<[^>]*>)?\s*\(", r1cs))
    called = set()
    for d, _, files in os.walk(ref):
        for f in files:
            if f.endswith(".rs") and "ristretto_point.rs" not in os.path.join(d, f).replace(os.path.join(ref, "src"), ""):
                text = re.sub(r"/\*.*?\*/", "", open(os.path.join(d, f)).read(), flags=re.S)
                text = re.sub(r"//[^\n]*", "", text)
                called |= set(re.findall(r"\b(?:cs|prover|verifier)\s*\.\s*([a-z_0-9]+)\s*\(", text))
    assert called and called <= defined, sorted(called - defined)
