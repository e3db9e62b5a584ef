"""CPU: host logic of the product's R1CS layer that needs no GPU -- the Merlin
transcript (known-answer vector) and circuit capture on the Verifier side
(multiplier / constraint counts of every gadget against SURVEY.md 8(a) and the
oracle's restatement)."""
import random

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
