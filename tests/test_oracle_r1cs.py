"""CPU: the oracle's R1CS restatement reproduces the reference's verdict table
(every positive and negative case of tests/edwards_point.rs, scalar.rs,
sk_know_gadget.rs), its proofs are reproducible under a seeded RNG, and tampering
with any proof field is rejected.  BASELINE config 1 lives here (CPU plumbing)."""
import random

import pytest

from oracle import r1cs as R
from oracle import sonny
from tests import ref_cases as rc
from tests.backends import OracleBackend

L = sonny.L


@pytest.fixture(scope="module")
def be(oracle):
    return OracleBackend(oracle)


@pytest.fixture(scope="module")
def rnd():
    return random.Random(2024)


def test_conditionally_select(be, rnd):
    """ref: tests/edwards_point.rs:97-111"""
    P1, ID = sonny.random_point(rnd), sonny.EdwardsPoint.identity()
    assert rc.cond_selection_roundtrip(be, P1, P1, 1)[0]
    assert not rc.cond_selection_roundtrip(be, P1, P1, 0)[0]
    assert rc.cond_selection_roundtrip(be, P1, ID, 0)[0]
    assert not rc.cond_selection_roundtrip(be, P1, ID, 1)[0]


def test_double_pub_points(be, rnd):
    """ref: tests/edwards_point.rs:175-189"""
    p, bad = sonny.random_point(rnd), sonny.random_point(rnd)
    assert rc.double_roundtrip(be, p, p.double())[0]
    assert not rc.double_roundtrip(be, p, bad)[0]


def test_add_pub_points(be, rnd):
    """ref: tests/edwards_point.rs:260-268 -- BASELINE config 1 (16 multipliers)"""
    p1, p2, bad = (sonny.random_point(rnd) for _ in range(3))
    ok, proof = rc.addition_roundtrip(be, p1, p2, p1 + p2)
    assert ok
    assert len(proof.ipp.L_vec) == 4 and len(proof.to_bytes()) == 1 + 11 * 32 + 8 * 32 + 64
    assert not rc.addition_roundtrip(be, p1, p2, bad)[0]


def test_point_committing(be, rnd):
    """ref: tests/edwards_point.rs:272-278"""
    A = sonny.random_point(rnd)
    assert rc.point_committing_roundtrip(be, A, A)[0]
    assert not rc.point_committing_roundtrip(be, A, A + A)[0]


def test_point_curve_eq(be, rnd):
    """ref: tests/edwards_point.rs:351-362"""
    assert rc.curve_eq_roundtrip(be, sonny.random_point(rnd))[0]
    assert not rc.curve_eq_roundtrip(be, sonny.EdwardsPoint(1, 1, 1, 1))[0]


def test_is_not_zero(be, rnd):
    """ref: src/gadgets/scalar.rs:77-83 (zero panics, :13-15)"""
    assert rc.is_not_zero_roundtrip(be, 1)[0]
    assert rc.is_not_zero_roundtrip(be, rnd.randrange(1, L))[0]
    with pytest.raises(RuntimeError):
        rc.is_not_zero_roundtrip(be, 0)


def test_sk_knowledge_small(be, rnd):
    """ref: sk_know_gadget.rs:110-124 with a 6-bit key (the 256-bit case runs on the GPU suite)"""
    basep = sonny.RISTRETTO_BASEPOINT
    sk = 0b101101
    pk = basep * sk
    assert rc.sk_knowledge_roundtrip(be, basep, pk, sk, nbits=6, cap=512)[0]
    assert not rc.sk_knowledge_roundtrip(be, basep, basep, sk, nbits=6, cap=512)[0]


def test_ristretto_add_is_134_multipliers(be, rnd):
    p1, p2 = sonny.RistrettoPoint(sonny.random_point(rnd)), sonny.RistrettoPoint(sonny.random_point(rnd))
    ok, proof = rc.ristretto_add_roundtrip(be, p1, p2, p1 + p2)
    assert ok and len(proof.ipp.L_vec) == 8          # 134 multipliers -> n+ = 256
    assert not rc.ristretto_add_roundtrip(be, p1, p2, p1 + p1)[0]


def test_scalar_mul_gadget(be, rnd):
    base = sonny.random_point(rnd)
    bits = [1, 0, 1, 1]            # little-endian: 13
    ok, proof = rc.scalar_mul_roundtrip(be, base, bits, base * 13)
    assert ok and len(proof.ipp.L_vec) == 7          # 27*4 + 4 = 112 multipliers -> 128
    assert not rc.scalar_mul_roundtrip(be, base, bits, base * 12)[0]
    assert not rc.scalar_mul_roundtrip(be, base, [1, 0, 2, 1], base * 13)[0]   # non-binary "bit"


def test_proofs_are_reproducible_and_rng_sensitive(be, rnd):
    p1, p2 = sonny.random_point(rnd), sonny.random_point(rnd)
    a = rc.addition_roundtrip(be, p1, p2, p1 + p2, rng=R.SeededRng(b"s1"))[1].to_bytes()
    b = rc.addition_roundtrip(be, p1, p2, p1 + p2, rng=R.SeededRng(b"s1"))[1].to_bytes()
    c = rc.addition_roundtrip(be, p1, p2, p1 + p2, rng=R.SeededRng(b"s2"))[1].to_bytes()
    assert a == b and a != c


def test_serialisation_roundtrip_and_tamper(be, rnd):
    p1, p2 = sonny.random_point(rnd), sonny.random_point(rnd)
    p3 = p1 + p2
    ok, proof = rc.addition_roundtrip(be, p1, p2, p3)
    raw = proof.to_bytes()
    assert R.R1CSProof.from_bytes(raw).to_bytes() == raw
    bp = be.bp_gens(128)

    def verify(pr):
        v = be.Verifier(be.Transcript(b"Point Addition"))
        g1, g2, g3 = (be.EdwardsGadget.from_point(p) for p in (p1, p2, p3))
        g3.equal(g1.add(g2, v), v)
        try:
            v.verify(pr, be.pc_gens, bp, bytes(32))
            return True
        except R.R1CSError:
            return False
    assert verify(R.R1CSProof.from_bytes(raw))
    # flip one bit in every 32-byte field in turn
    for off in range(1, len(raw), 32):
        bad = bytearray(raw)
        bad[off] ^= 1
        try:
            pr = R.R1CSProof.from_bytes(bytes(bad))
        except R.R1CSError:
            continue
        assert not verify(pr), off
    # wrong transcript label
    v = be.Verifier(be.Transcript(b"Point Addition?"))
    g1, g2, g3 = (be.EdwardsGadget.from_point(p) for p in (p1, p2, p3))
    g3.equal(g1.add(g2, v), v)
    with pytest.raises(R.VerificationError):
        v.verify(proof, be.pc_gens, bp, bytes(32))


def test_generators_too_short(be, rnd):
    p = sonny.random_point(rnd)
    with pytest.raises(R.InvalidGeneratorsLength):
        rc.double_roundtrip(be, p, p.double(), cap=8)


def test_multiplier_counts_match_survey(be, rnd):
    """add 12, double 10, equal 4, select 4, curve-eq 8, from_point 40, scalar_mul 27/bit, sk 29/bit"""
    from oracle import gadgets as G
    p = sonny.random_point(rnd)

    def count(f):
        v = be.Verifier(be.Transcript(b"count"))
        f(v)
        return v.num_vars, len(v.constraints)
    E = G.SonnyEdwardsPointGadget
    g = E.from_point(p)
    assert count(lambda cs: g.add(g, cs)) == (12, 30)
    assert count(lambda cs: g.double(cs)) == (10, 24)
    assert count(lambda cs: g.equal(g, cs)) == (4, 10)
    assert count(lambda cs: g.satisfy_curve_eq(cs)) == (8, 19)
    assert count(lambda cs: g.conditionally_select(R.LC.of(1), cs)) == (4, 10)
    assert count(lambda cs: G.SonnyRistrettoPointGadget.from_point(sonny.RistrettoPoint(p), cs))[0] == 40
    v = be.Verifier(be.Transcript(b"count"))
    bits = [v.commit(bytes(32)) for _ in range(3)]
    E.scalar_mul(g, bits, v)
    assert v.num_vars == 27 * 3


def test_randomized_constraints_shuffle(be, rnd):
    """SURVEY 8f4 (two-phase constraints; parity unpinned: the reference registers none, so it holds no vector).
    Checks the protocol's own properties: completeness, soundness on a non-permutation, the version-1 encoding."""
    xs = [rnd.randrange(L) for _ in range(5)]
    perm = xs[:]
    rnd.shuffle(perm)
    ok, proof = rc.shuffle_roundtrip(be, xs, perm)
    assert ok
    raw = proof.to_bytes()
    assert raw[0] == 1 and len(raw) == 1 + 14 * 32 + 2 * 3 * 32 + 64      # two-phase: A_I2, A_O2, S2 present; n = 8
    assert R.R1CSProof.from_bytes(raw).to_bytes() == raw
    bad = perm[:]
    bad[2] = (bad[2] + 1) % L
    assert not rc.shuffle_roundtrip(be, xs, bad)[0]
    # both phases populated: 3 first-phase multipliers, 2 * (k - 1) second-phase ones
    ok, proof = rc.shuffle_roundtrip(be, xs, perm, first_phase_squares=3)
    assert ok and proof.to_bytes()[0] == 1
    assert not rc.shuffle_roundtrip(be, xs, bad, first_phase_squares=3)[0]
    # k = 1 registers no callback: a one-phase proof
    ok, proof = rc.shuffle_roundtrip(be, xs[:1], xs[:1])
    assert ok and proof.to_bytes()[0] == 0
    # k = 2: one multiplier per side
    assert rc.shuffle_roundtrip(be, xs[:2], xs[1::-1])[0]


def test_challenge_scalar_outside_the_randomization_phase(be):
    p = be.Prover(be.Transcript(b"x"))
    with pytest.raises(R.R1CSError):
        p.challenge_scalar(b"c")


def test_circuits_without_multipliers(be):
    """n = 0 pads to one multiplier slot and an inner-product argument of zero rounds (bulletproofs: padded_n = 1)"""
    ok, proof = rc.linear_only_roundtrip(be, 5, 7, 12)
    assert ok and len(proof.ipp.L_vec) == 0
    assert not rc.linear_only_roundtrip(be, 5, 7, 13)[0]
    ok, proof = rc.linear_only_roundtrip(be, 5, 7, 12, n_mult=1)
    assert ok and len(proof.ipp.L_vec) == 0
    assert rc.linear_only_roundtrip(be, 5, 7, 12, n_mult=3)[0]
