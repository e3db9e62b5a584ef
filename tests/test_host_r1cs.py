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

    # from_lcs: the Ristretto gadget over four linear combinations, no point assignment (ref: ristretto_point.rs:34-45)
    coords = [p.X, p.Y, p.Z, p.T]
    assert _count(lambda cs: g.SonnyRistrettoPointGadget.from_lcs(gp, 4, cs)) == \
        oracle_count(lambda cs: OG.SonnyRistrettoPointGadget.from_lcs(coords, cs))

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


def _cond_select_terms_from_the_rust(X, Y, Z, T):
    """The (kind, index, coeff, constraint) list that reference/src/gadgets/point/edwards_point.rs:215-253
    (ristretto_point.rs:184-226 is the same text) produces for `point.conditionally_select(bit, cs)` with a
    constant point and bit = Variable::Committed(0), derived by hand from the Rust, multiply by multiply:
      :221  (_, bit, x')      = multiply(X, bit)            multiplier 0; `bit` := MultiplierRight(0)
      :227  (bit_y, _, bit*y) = multiply(bit, Y)            multiplier 1; bit_y shadows `bit` inside the block only
      :229  constrain(y' - bit*y - 1 + bit_y),  y' = bit*y + 1 - bit_y
      :237  (bit_z, _, bit*z) = multiply(bit, Z)            multiplier 2; left input is MultiplierRight(0) again
      :239  constrain(z' - bit*z - 1 + bit_z)
      :246  (_, _, t')        = multiply(T, bit)            multiplier 3; right input is MultiplierRight(0)
    Each multiply(l, r) appends two constraints: l - MultiplierLeft(i) and r - MultiplierRight(i)."""
    C, Lf, R, O, ONE = 0, 1, 2, 3, 4
    m1 = g.L - 1
    return [
        (ONE, 0, X, 0), (Lf, 0, m1, 0),          # X - L0
        (C, 0, 1, 1), (R, 0, m1, 1),             # bit - R0
        (R, 0, 1, 2), (Lf, 1, m1, 2),            # R0 - L1
        (ONE, 0, Y, 3), (R, 1, m1, 3),           # Y - R1
        (O, 1, 1, 4), (ONE, 0, 1, 4), (Lf, 1, m1, 4), (O, 1, m1, 4), (ONE, 0, m1, 4), (Lf, 1, 1, 4),
        (R, 0, 1, 5), (Lf, 2, m1, 5),            # R0 - L2   (not L1: the inner `bit` went out of scope at :231)
        (ONE, 0, Z, 6), (R, 2, m1, 6),           # Z - R2
        (O, 2, 1, 7), (ONE, 0, 1, 7), (Lf, 2, m1, 7), (O, 2, m1, 7), (ONE, 0, m1, 7), (Lf, 2, 1, 7),
        (ONE, 0, T, 8), (Lf, 3, m1, 8),          # T - L3
        (R, 0, 1, 9), (R, 3, m1, 9),             # R0 - R3   (not L2)
    ]


def test_conditionally_select_wiring_matches_the_rust(oracle):
    """block-scoped shadowing of `bit` -- the wiring of every circuit that uses conditionally_select,
    scalar_mul or sk_knowledge.  Product (both gadget types) and oracle against the hand-derived list."""
    from oracle import gadgets as OG
    from tests.backends import OracleBackend
    rnd = random.Random(11)
    p = sonny.random_point(rnd)
    want = _cond_select_terms_from_the_rust(p.X % g.L, p.Y % g.L, p.Z % g.L, p.T % g.L)
    for make in (lambda: g.SonnyEdwardsPointGadget.from_point(p),):
        v = g.Verifier(b"wiring")
        bit = v.commit(bytes(32))
        make().conditionally_select(g.LC.of(bit), v)
        assert v.export_terms() == want
    # the Ristretto gadget's conditionally_select on the same LCs (ristretto_point.rs:184-226)
    v = g.Verifier(b"wiring")
    bit = v.commit(bytes(32))
    import ctypes
    from bulletproofs_gadgets_amd._native import lib, check
    e = g.SonnyEdwardsPointGadget.from_point(p)
    vv, cc, n = g.LC.of(bit).marshal()
    h = ctypes.c_void_p()
    check(lib().bpg_ristretto_conditionally_select(v._cs, e._h, vv, cc, n, ctypes.byref(h)))
    lib().bpg_point_free(h)
    assert v.export_terms() == want
    # the oracle's restatement
    be = OracleBackend(oracle)
    ov = be.Verifier(be.Transcript(b"wiring"))
    obit = ov.commit(bytes(32))
    OG.SonnyEdwardsPointGadget.from_point(p).conditionally_select(be.lc(obit), ov)
    kind = {"C": 0, "L": 1, "R": 2, "O": 3, "One": 4}
    got = [(kind[var.kind], var.i, c % g.L, q) for q, lc in enumerate(ov.constraints) for var, c in lc.terms]
    assert got == want


def test_foreign_variables_are_rejected():
    """a variable that this constraint system never issued (another prover's, a stale handle, a
    made-up kind) is BPG_EINVAL at the ABI, not a write past the flattening plan"""
    from bulletproofs_gadgets_amd import BpgError, _native
    v = g.Verifier(b"vars")
    c0 = v.commit(bytes(32))
    l, r, o = v.multiply(g.LC.of(c0), g.LC.of(1))
    for bad in (g.Variable((0 << 32) | 1),          # Committed(1): only one commitment exists
                g.Variable((1 << 32) | 1),          # MultiplierLeft(1): only multiplier 0 exists
                g.Variable((3 << 32) | 0x0fffffff),
                g.Variable((5 << 32) | 0),          # no such kind
                g.Variable((4 << 32) | 7)):         # One carries no index
        for call in (lambda: v.constrain(g.LC.of(bad)), lambda: v.multiply(g.LC.of(bad), g.LC.of(1)),
                     lambda: v.multiply(g.LC.of(c0), g.LC.of(bad)), lambda: g.binary_constrain_gadget(v, bad)):
            with pytest.raises(BpgError) as e:
                call()
            assert e.value.code == _native.EINVAL
    # a gadget assembled from foreign variables is caught when it reaches the constraint system
    gp = g.SonnyEdwardsPointGadget.from_vars([g.Variable((1 << 32) | 9)] * 4)
    with pytest.raises(BpgError) as e:
        gp.double(v)
    assert e.value.code == _native.EINVAL
    assert v.stats()[0] >= 1        # the system is still usable for well-formed input
    v.constrain(g.LC.of(o) - g.LC.of(l))
