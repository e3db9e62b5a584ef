"""CPU: host logic of the product's R1CS layer that needs no GPU -- the Merlin
transcript (known-answer vector) and circuit capture on the Verifier side
(multiplier / constraint counts of every gadget against SURVEY.md 8(a) and the
oracle's restatement)."""
import random

import pytest

from bulletproofs_gadgets_amd import r1cs as g
from oracle import sonny


def test_merlin_known_answer():
    out = g.transcript_kat(b"test protocol", b"some label", b"some data", b"challenge", 32)
    assert out.hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"


def _count(f):
    v = g.Verifier(b"count")
    f(v)
    return v.stats()


def test_gadget_shapes_match_survey_and_oracle(oracle):
    from oracle import gadgets as OG
    from tests.backends import OracleBackend
    be = OracleBackend(oracle)
    rnd = random.Random(3)
    p = sonny.random_point(rnd)
    E = g.SonnyEdwardsPointGadget
    gp = E.from_point(p)
    assert _count(lambda cs: gp.add(gp, cs)) == (12, 30)
    assert _count(lambda cs: gp.double(cs)) == (10, 24)
    assert _count(lambda cs: gp.equal(gp, cs)) == (4, 10)
    assert _count(lambda cs: gp.satisfy_curve_eq(cs)) == (8, 19)
    assert _count(lambda cs: gp.conditionally_select(g.LC.of(1), cs)) == (4, 10)
    rp = sonny.RistrettoPoint(p)
    n_from_point = _count(lambda cs: g.SonnyRistrettoPointGadget.from_point(rp, cs))
    assert n_from_point[0] == 40

    def oracle_count(f):
        v = be.Verifier(be.Transcript(b"count"))
        f(v)
        return v.num_vars, len(v.constraints)
    assert n_from_point == oracle_count(lambda cs: OG.SonnyRistrettoPointGadget.from_point(rp, cs))

    def smul(cs, EG):
        bits = [cs.commit(bytes(32)) for _ in range(5)]
        EG.scalar_mul(EG.from_point(p), bits, cs)
    assert _count(lambda cs: smul(cs, E)) == oracle_count(lambda cs: smul(cs, OG.SonnyEdwardsPointGadget))
    assert _count(lambda cs: smul(cs, E))[0] == 27 * 5

    def skk(cs, RG, fn):
        bits = [cs.commit(bytes(32)) for _ in range(3)]
        fn(cs, RG.from_point(rp, cs), RG.from_point(rp, cs), bits)
    a = _count(lambda cs: skk(cs, g.SonnyRistrettoPointGadget, g.sk_knowledge_gadget))
    assert a == oracle_count(lambda cs: skk(cs, OG.SonnyRistrettoPointGadget, OG.sk_knowledge_gadget))
    assert a[0] == 2 * 40 + 29 * 3 + 2


def test_hard_errors_mirror_the_reference_panics():
    import pytest
    from bulletproofs_gadgets_amd import BpgError, _native
    v = g.Verifier(b"x")
    with pytest.raises(BpgError) as e:           # assert_eq!(commitments.len(), 4), edwards_point.rs:282
        g.SonnyEdwardsPointGadget.verifier_commit_to_sonny_edwards_point(v, [bytes(32)] * 3)
    assert e.value.code == _native.EGADGET
    gp = g.SonnyEdwardsPointGadget.from_point(sonny.BASEPOINT)
    with pytest.raises(BpgError) as e:           # assert!(lcs.len() == 4), ristretto_point.rs:35
        g.SonnyRistrettoPointGadget.from_lcs(gp, 3, v)
    assert e.value.code == _native.EGADGET
    with pytest.raises(RuntimeError):            # panic!("Attempting to divide by 0 ..."), scalar.rs:13-15
        g.nonzero_gadget(g.LC.of(0), 0, v)


def test_host_keccak_bodies_and_shake(host_hooks):
    """the generated Keccak-f (baseline body and the BMI body picked at run time) against hashlib,
    including squeezes that cross many rate boundaries (the generator chain reads megabytes)"""
    import ctypes
    import hashlib
    import random
    rnd = random.Random(3)
    for n_in, n_out in ((0, 32), (5, 64), (135, 136), (136, 137), (137, 1000), (1000, 5000)):
        msg = bytes(rnd.randrange(256) for _ in range(n_in))
        out = ctypes.create_string_buffer(n_out)
        host_hooks.hh_shake256(out, ctypes.c_size_t(n_out), msg, ctypes.c_size_t(n_in))
        assert out.raw == hashlib.shake_256(msg).digest(n_out)
    for _ in range(20):
        st = [rnd.getrandbits(64) for _ in range(25)]
        a = (ctypes.c_uint64 * 25)(*st)
        b = (ctypes.c_uint64 * 25)(*st)
        host_hooks.hh_keccak_f(a, 0)
        host_hooks.hh_keccak_f(b, 1)
        assert list(a) == list(b)


def test_host_transcript_rng_matches_oracle(host_hooks, oracle):
    """the prover's blinding stream (merlin TranscriptRng over STROBE) equals the oracle's, scalar for scalar"""
    import ctypes
    from oracle import r1cs as R
    n = 300                      # crosses the STROBE rate many times
    wit, ext = bytes(range(31)) + b"\x05", bytes(range(100, 132))      # witness below l
    out = ctypes.create_string_buffer(32 * n)
    host_hooks.hh_transcript_rng(out, ctypes.c_size_t(n), b"rng-test", ctypes.c_size_t(8), wit, ext)
    tr = R.Transcript(oracle, b"rng-test")
    rng = tr.build_rng([int.from_bytes(wit, "little")], ext)          # rekeys with label "v_blinding"
    want = b"".join(x.to_bytes(32, "little") for x in rng.scalars(n))
    assert out.raw == want


def test_host_keccak_x8_and_rng_lanes(host_hooks, oracle):
    """eight interleaved Keccak states (AVX-512 body where the CPU has it) against the one-state body, and eight
    blinding RNGs drawn in lock-step against each RNG drawn on its own (the oracle's stream), including the
    hand-back to the single-state RNG for the draws that follow"""
    import ctypes
    import random
    from oracle import r1cs as R
    rnd = random.Random(8)
    for _ in range(10):
        st = [rnd.getrandbits(64) for _ in range(200)]
        a = (ctypes.c_uint64 * 200)(*st)
        b = (ctypes.c_uint64 * 200)(*st)
        host_hooks.hh_keccak_x8(a, 0)
        host_hooks.hh_keccak_x8(b, 1)
        assert list(a) == list(b)
        one = (ctypes.c_uint64 * 25)(*[st[8 * i + 3] for i in range(25)])
        host_hooks.hh_keccak_f(one, 0)
        assert list(one) == [a[8 * i + 3] for i in range(25)]
    n, tail = 40, 5
    wits = [bytes(rnd.randrange(256) for _ in range(31)) + b"\x03" for _ in range(8)]
    exts = [bytes(rnd.randrange(256) for _ in range(32)) for _ in range(8)]
    out = ctypes.create_string_buffer(32 * (n + tail) * 8)
    host_hooks.hh_transcript_rng_lanes.restype = ctypes.c_int
    ok = host_hooks.hh_transcript_rng_lanes(out, ctypes.c_size_t(n), ctypes.c_size_t(tail), b"lanes-test", ctypes.c_size_t(10),
                                            b"".join(wits), b"".join(exts))
    assert ok == 1
    for lane in range(8):
        tr = R.Transcript(oracle, b"lanes-test")
        rng = tr.build_rng([int.from_bytes(wits[lane], "little")], exts[lane])
        want = b"".join(x.to_bytes(32, "little") for x in rng.scalars(n + tail))
        assert out.raw[32 * (n + tail) * lane:32 * (n + tail) * (lane + 1)] == want, lane


def test_termvec_under_sanitizers(tmp_path):
    """LinearCombination's inline small-vector (r1cs.h: TermVec), host code, built with ASan + UBSan"""
    import os
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "termvec_test")
    src = os.path.join(root, "tests", "native", "termvec_test.hip")
    subprocess.run([hipcc, "-O1", "-g", "-std=c++17", "--offload-arch=gfx950", "-Xarch_host", "-fsanitize=address,undefined",
                    "-Xarch_host", "-fno-omit-frame-pointer", "-I" + os.path.join(root, "include"), src, "-o", exe], check=True, cwd=root)
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0 and "termvec: ok" in r.stdout, r.stdout + r.stderr
