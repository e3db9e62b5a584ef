"""GPU: randomized (two-phase) constraints through the C ABI -- SURVEY 8f4.

The reference turns the feature on (Cargo.toml:18) and registers no randomized constraint, so there is no reference
vector for this path: parity unpinned.  What is checked: proof BYTES equal the oracle's under the same seeded RNG
(the oracle restates bulletproofs' two-phase prover / verifier), each side verifies the other's proofs, and the
protocol's own properties (soundness on a non-permutation, tampering with the second-phase commitments)."""
import random

import pytest

from oracle import r1cs as R
from oracle import sonny
from tests import ref_cases as rc
from tests.backends import GpuBackend, OracleBackend

pytestmark = pytest.mark.gpu
L = sonny.L


@pytest.fixture(scope="module")
def gbe(gpu_ctx):
    return GpuBackend(gpu_ctx)


@pytest.fixture(scope="module")
def obe(oracle):
    return OracleBackend(oracle)


@pytest.fixture()
def rnd():
    return random.Random(4242)


def _both(gbe, obe, *args, seed=b"2p", **kw):
    ok_g, pg = rc.shuffle_roundtrip(gbe, *args, rng=R.SeededRng(seed), **kw)
    ok_o, po = rc.shuffle_roundtrip(obe, *args, rng=R.SeededRng(seed), **kw)
    assert ok_g == ok_o
    assert pg.to_bytes() == po.to_bytes()
    return ok_g, pg


@pytest.mark.parametrize("k,squares", [(2, 0), (5, 0), (5, 3), (8, 1), (17, 2), (33, 0)])
def test_shuffle_bytes_equal_the_oracle(gbe, obe, rnd, k, squares):
    """second phase only (first phase empty: A_I1 = i_bl1 * B~), both phases, padded and exact powers of two"""
    xs = [rnd.randrange(L) for _ in range(k)]
    perm = xs[:]
    rnd.shuffle(perm)
    ok, proof = _both(gbe, obe, xs, perm, first_phase_squares=squares, cap=128)
    assert ok
    raw = proof.to_bytes()
    assert raw[0] == 1                       # A_I2, A_O2, S2 on the wire
    bad = perm[:]
    bad[k // 2] = (bad[k // 2] + 1) % L
    assert not _both(gbe, obe, xs, bad, first_phase_squares=squares, cap=128)[0]


def test_single_element_shuffle_is_a_one_phase_proof(gbe, obe, rnd):
    x = [rnd.randrange(L)]
    ok, proof = _both(gbe, obe, x, x)
    assert ok and proof.to_bytes()[0] == 0


def _verify(be, comms, k, proof, bp):
    v = be.Verifier(be.Transcript(b"ShuffleProof"))
    vv = [v.commit(c) for c in comms]
    rc.shuffle_gadget(v, vv[:k], vv[k:])
    return v.verify(proof, be.pc_gens, bp, bytes(32))


def _prove(be, xs, ys, seed, cap=64):
    rng = R.SeededRng(seed)
    bp = be.bp_gens(cap)
    p = be.Prover(be.Transcript(b"ShuffleProof"))
    comms, vars_ = [], []
    for v in xs + ys:
        V, var = p.commit(v, rng.scalar())
        comms.append(V)
        vars_.append(var)
    rc.shuffle_gadget(p, vars_[:len(xs)], vars_[len(xs):])
    return p, comms, bp, rng


def test_cross_verification_and_tampering(gbe, obe, rnd):
    k = 6
    xs = [rnd.randrange(L) for _ in range(k)]
    ys = xs[::-1]
    pg, comms, bpg_, rng = _prove(gbe, xs, ys, b"g")
    proof_g = pg.prove(bpg_, rng.fill_bytes(32))
    po, comms_o, bpo, rng = _prove(obe, xs, ys, b"o")
    proof_o = po.prove(bpo, rng.fill_bytes(32))
    assert _verify(obe, comms, k, R.R1CSProof.from_bytes(proof_g.to_bytes()), bpo)
    assert _verify(gbe, comms_o, k, proof_o, bpg_)
    raw = bytearray(proof_g.to_bytes())
    other = obe.pc_gens.commit(5, 7)               # some other valid encoding
    for field in (4, 5, 6):                  # A_I2, A_O2, S2: byte 0 is the version, then 32-byte fields
        t = bytearray(raw)
        t[1 + 32 * (field - 1):1 + 32 * field] = other
        with pytest.raises(gbe.VerificationError):
            _verify(gbe, comms, k, bytes(t), bpg_)
    # a verifier that does not register the randomized constraints reads another transcript
    v = gbe.Verifier(b"ShuffleProof")
    for c in comms:
        v.commit(c)
    with pytest.raises(gbe.VerificationError):
        v.verify(bytes(raw), None, bpg_, bytes(32))


def test_prove_many_and_verify_batch_with_two_phase_proofs(gbe, rnd):
    """provers with randomized constraints are proved one at a time inside prove_many; the bytes are those of
    separate prove() calls; verify_batch takes one- and two-phase proofs together"""
    g = gbe.g
    k = 4
    xs = [rnd.randrange(L) for _ in range(k)]
    ys = xs[1:] + xs[:1]
    ext = bytes(range(96))
    provers, singles = [], []
    for rep in range(2):
        ps = []
        for j in range(3):
            kk = 1 if j == 1 else k            # the middle one: k = 1, a one-phase proof
            p, comms, bp, _ = _prove(gbe, xs[:kk], ys[:kk] if kk > 1 else xs[:1], b"pm%d" % j)
            ps.append((p, comms, kk))
        (provers if rep == 0 else singles).extend(ps)
    many = g.Prover.prove_many([p for p, _, _ in provers], ext)
    for j, (p, _, _) in enumerate(singles):
        assert p.prove(None, ext[32 * j:32 * j + 32]).to_bytes() == many[j].to_bytes()
    assert [pr.to_bytes()[0] for pr in many] == [1, 0, 1]

    def verifiers():
        out = []
        for _, comms, kk in provers:
            v = gbe.Verifier(b"ShuffleProof")
            vv = [v.commit(c) for c in comms]
            rc.shuffle_gadget(v, vv[:kk], vv[kk:])
            out.append(v)
        return out
    assert g.verify_batch(verifiers(), many, bp)
    t = bytearray(many[2].to_bytes())
    t[1 + 32 * 3 + 5] ^= 1                     # inside A_I2
    with pytest.raises(g.VerificationError):
        g.verify_batch(verifiers(), [many[0], many[1], bytes(t)], bp)


def test_callback_errors_surface(gbe, rnd):
    gbe.bp_gens(64)
    p = gbe.Prover(b"err")
    _, x = p.commit(3, 5)

    class Boom(Exception):
        pass

    def cb(rcs):
        rcs.challenge_scalar(b"c")
        raise Boom()
    p.specify_randomized_constraints(cb)
    with pytest.raises(Boom):
        p.prove(None, bytes(32))
    # challenge_scalar is a second-phase call
    q = gbe.Prover(b"err")
    from bulletproofs_gadgets_amd import r1cs as g
    with pytest.raises(RuntimeError):
        g._RandomizedCS(q._cs).challenge_scalar(b"c")
    # the second phase may outgrow the generators: InvalidGeneratorsLength, as in the one-phase path
    gbe.bp_gens(4)
    r = gbe.Prover(b"big")
    xs = [rnd.randrange(L) for _ in range(6)]
    vars_ = [r.commit(v, 1)[1] for v in xs + xs]
    rc.shuffle_gadget(r, vars_[:6], vars_[6:])          # 10 second-phase multipliers > 4
    with pytest.raises(RuntimeError, match="BPG_EGENS"):
        r.prove(None, bytes(32))


def test_nested_registration_is_refused_at_the_c_abi(gbe):
    """upstream's types rule out specify_randomized_constraints from inside a callback; here the rcs handle
    is the same object, so the C ABI must answer BPG_EINVAL (not queue a callback that never runs).  The Python
    mirror refuses on its own, so the call goes to the library directly."""
    import ctypes
    from bulletproofs_gadgets_amd import _native
    lib = _native.lib()
    bp = gbe.bp_gens(16)
    seen = {}
    inner = _native.RANDOMIZED_FN(lambda _u, _rcs: 0)

    def outer(_user, rcs_ptr):
        seen["rc"] = lib.bpg_cs_specify_randomized_constraints(rcs_ptr, inner, None)
        return 0
    fn = _native.RANDOMIZED_FN(outer)
    for side in ("prover", "verifier"):
        seen.clear()
        p = gbe.Prover(b"nested")
        V, _ = p.commit(3, 5)
        assert lib.bpg_cs_specify_randomized_constraints(p._cs, fn, None) == 0
        proof = p.prove(None, bytes(32))
        if side == "verifier":
            seen.clear()
            v = gbe.Verifier(b"nested")
            v.commit(V)
            assert lib.bpg_cs_specify_randomized_constraints(v._cs, fn, None) == 0
            v.verify(proof, None, bp, bytes(32))
        assert seen["rc"] == _native.EINVAL


def test_two_phase_prover_proves_once(gbe, rnd):
    """a prove() that fails after the callbacks have run leaves no half-initialised state behind: the
    prover is consumed (upstream: prove(self)) and says so."""
    import bulletproofs_gadgets_amd as bpg
    from bulletproofs_gadgets_amd import _native
    gbe.bp_gens(4)
    r = gbe.Prover(b"big")
    xs = [rnd.randrange(L) for _ in range(6)]
    vars_ = [r.commit(v, 1)[1] for v in xs + xs]
    rc.shuffle_gadget(r, vars_[:6], vars_[6:])
    with pytest.raises(RuntimeError, match="BPG_EGENS"):
        r.prove(None, bytes(32))
    with pytest.raises(bpg.BpgError) as e:
        r.prove(None, bytes(32))
    assert e.value.code == _native.EINVAL


def test_early_blinding_draws_give_way_to_randomized_constraints(gbe, rnd):
    """bpg_prover_begin_blindings on a prover that then registers randomized constraints: the helper is stopped (a
    two-phase proof draws in two rounds) and the proof is the one prove() gives with that external randomness."""
    xs = [rnd.randrange(1, R.L) for _ in range(4)]
    ys = list(reversed(xs))
    ext = bytes(range(32))
    p0, _, bp, _ = _prove(gbe, xs, ys, b"eb")
    want = p0.prove(bp, ext).to_bytes()

    rng = R.SeededRng(b"eb")
    p = gbe.Prover(gbe.Transcript(b"ShuffleProof"))
    vars_ = []
    for v in xs + ys:
        _, var = p.commit(v, rng.scalar())
        vars_.append(var)
    p.begin_blindings(ext)                             # before the gadget registers its randomized constraints
    rc.shuffle_gadget(p, vars_[:4], vars_[4:])
    assert p.prove(bp, bytes(32)).to_bytes() == want      # prove's own randomness is not asked
