"""The reference's own test programs (tests/edwards_point.rs, src/gadgets/scalar.rs,
src/gadgets/sk_knowledge/sk_know_gadget.rs), restated once over an abstract backend
so that the SAME cases run against the oracle (CPU) and against the product (GPU,
through the C ABI).  A backend `be` provides:

    be.Transcript(label) / be.Prover(transcript) / be.Verifier(transcript)
    be.pc_gens / be.bp_gens(capacity)
    be.EdwardsGadget / be.RistrettoGadget / be.gadgets (binary/nonzero/sk_knowledge)
    be.VerificationError

Every helper returns True for Ok(()) and False for Err(R1CSError::VerificationError),
like `.is_ok()` / `.is_err()` in the reference.  RNG: the reference draws from
thread_rng; here every draw comes from a seeded stream so proofs are reproducible.
"""
from oracle import sonny
from oracle.r1cs import SeededRng


def _ok(fn, be):
    try:
        fn()
        return True
    except be.VerificationError:
        return False


# ---- tests/edwards_point.rs:18-95  conditional selection
def cond_selection_roundtrip(be, P1, P2, selector, rng=None, cap=64):
    rng = rng or SeededRng(b"cond")
    bp = be.bp_gens(cap)
    tr = be.Transcript(b"Conditional_Selection!")
    prover = be.Prover(tr)
    comm, cond = prover.commit(selector, rng.scalar())
    g1 = be.EdwardsGadget.from_point(P1)
    g2 = be.EdwardsGadget(P2.X, P2.Y, P2.Z, P2.T)
    g1 = g1.conditionally_select(be.lc(cond), prover)
    g1.equal(g2, prover)
    proof = prover.prove(bp, rng.fill_bytes(32))

    def verify():
        tr2 = be.Transcript(b"Conditional_Selection!")
        v = be.Verifier(tr2)
        c = v.commit(comm)
        h1 = be.EdwardsGadget.from_point(P1)
        h2 = be.EdwardsGadget(P2.X, P2.Y, P2.Z, P2.T)
        h1 = h1.conditionally_select(be.lc(c), v)
        h1.equal(h2, v)
        v.verify(proof, be.pc_gens, bp, rng.fill_bytes(32))
    return _ok(verify, be), proof


# ---- tests/edwards_point.rs:115-173  doubling with public points
def double_roundtrip(be, point, point_doubled, rng=None, cap=128):
    rng = rng or SeededRng(b"dbl")
    bp = be.bp_gens(cap)

    def circuit(cs):
        g = be.EdwardsGadget.from_point(point)
        g2 = g.double(cs)
        g2.equal(be.EdwardsGadget.from_point(point_doubled), cs)
    prover = be.Prover(be.Transcript(b"IsRistretto?"))
    circuit(prover)
    proof = prover.prove(bp, rng.fill_bytes(32))

    def verify():
        v = be.Verifier(be.Transcript(b"IsRistretto?"))
        circuit(v)
        v.verify(proof, be.pc_gens, bp, rng.fill_bytes(32))
    return _ok(verify, be), proof


# ---- tests/edwards_point.rs:193-258  addition with public points (BASELINE config 1)
def addition_roundtrip(be, p1, p2, p3, rng=None, cap=128):
    rng = rng or SeededRng(b"add")
    bp = be.bp_gens(cap)

    def circuit(cs):
        g1, g2, g3 = (be.EdwardsGadget.from_point(p) for p in (p1, p2, p3))
        s = g1.add(g2, cs)
        g3.equal(s, cs)
    prover = be.Prover(be.Transcript(b"Point Addition"))
    circuit(prover)
    proof = prover.prove(bp, rng.fill_bytes(32))

    def verify():
        v = be.Verifier(be.Transcript(b"Point Addition"))
        circuit(v)
        v.verify(proof, be.pc_gens, bp, rng.fill_bytes(32))
    return _ok(verify, be), proof


# ---- tests/edwards_point.rs:280-347  committed points
def point_committing_roundtrip(be, p1, p2, rng=None, cap=128):
    rng = rng or SeededRng(b"commit")
    bp = be.bp_gens(cap)
    prover = be.Prover(be.Transcript(b"PointDouble"))
    g1, c1 = be.EdwardsGadget.prover_commit_to_sonny_edwards_point(prover, p1, rng)
    g2, c2 = be.EdwardsGadget.prover_commit_to_sonny_edwards_point(prover, p2, rng)
    commitments = c1 + c2
    g1.equal(g2, prover)
    proof = prover.prove(bp, rng.fill_bytes(32))

    def verify():
        v = be.Verifier(be.Transcript(b"PointDouble"))
        h1 = be.EdwardsGadget.verifier_commit_to_sonny_edwards_point(v, commitments[0:4])
        h2 = be.EdwardsGadget.verifier_commit_to_sonny_edwards_point(v, commitments[4:8])
        h1.equal(h2, v)
        v.verify(proof, be.pc_gens, bp, rng.fill_bytes(32))
    return _ok(verify, be), proof


# ---- tests/edwards_point.rs:364-414  curve equation
def curve_eq_roundtrip(be, p, rng=None, cap=128):
    rng = rng or SeededRng(b"curve")
    bp = be.bp_gens(cap)

    def circuit(cs):
        be.EdwardsGadget.from_point(p).satisfy_curve_eq(cs)
    prover = be.Prover(be.Transcript(b"PointDouble"))
    circuit(prover)
    proof = prover.prove(bp, rng.fill_bytes(32))

    def verify():
        v = be.Verifier(be.Transcript(b"PointDouble"))
        circuit(v)
        v.verify(proof, be.pc_gens, bp, rng.fill_bytes(32))
    return _ok(verify, be), proof


# ---- src/gadgets/scalar.rs:33-75  is-nonzero
def is_not_zero_roundtrip(be, fe, rng=None, cap=32):
    rng = rng or SeededRng(b"nz")
    bp = be.bp_gens(cap)

    def circuit(cs):
        be.gadgets.nonzero_gadget(be.lc(fe), fe, cs)
    prover = be.Prover(be.Transcript(b"Is zero?"))
    circuit(prover)
    proof = prover.prove(bp, rng.fill_bytes(32))

    def verify():
        v = be.Verifier(be.Transcript(b"Is zero?"))
        circuit(v)
        v.verify(proof, be.pc_gens, bp, rng.fill_bytes(32))
    return _ok(verify, be), proof


# ---- src/gadgets/sk_knowledge/sk_know_gadget.rs:40-105  secret-key knowledge
def sk_knowledge_roundtrip(be, basep, pk, sk, nbits=256, rng=None, cap=8192):
    rng = rng or SeededRng(b"sk")
    bp = be.bp_gens(cap)
    bits = sonny.scalar_bits(sk, nbits)
    prover = be.Prover(be.Transcript(b"Sk_knowledge"))
    comms, vars_ = [], []
    for b in bits:
        V, var = prover.commit(b, rng.scalar())
        comms.append(V)
        vars_.append(var)
    bg = be.RistrettoGadget.from_point(basep, prover)
    pg = be.RistrettoGadget.from_point(pk, prover)
    be.gadgets.sk_knowledge_gadget(prover, bg, pg, vars_)
    proof = prover.prove(bp, rng.fill_bytes(32))

    def verify():
        v = be.Verifier(be.Transcript(b"Sk_knowledge"))
        vv = [v.commit(c) for c in comms]
        b2 = be.RistrettoGadget.from_point(basep, v)
        p2 = be.RistrettoGadget.from_point(pk, v)
        be.gadgets.sk_knowledge_gadget(v, b2, p2, vv)
        v.verify(proof, be.pc_gens, bp, rng.fill_bytes(32))
    return _ok(verify, be), proof


# ---- BASELINE config 5 unit: Ristretto gadget P1 + P2 = P3 (134 multipliers)
def ristretto_add_roundtrip(be, p1, p2, p3, rng=None, cap=256):
    rng = rng or SeededRng(b"radd")
    bp = be.bp_gens(cap)

    def circuit(cs):
        g1, g2, g3 = (be.RistrettoGadget.from_point(p, cs) for p in (p1, p2, p3))
        s = g1.add(cs, g2)
        g3.equals(cs, s)
    prover = be.Prover(be.Transcript(b"RistrettoAdd"))
    circuit(prover)
    proof = prover.prove(bp, rng.fill_bytes(32))

    def verify():
        v = be.Verifier(be.Transcript(b"RistrettoAdd"))
        circuit(v)
        v.verify(proof, be.pc_gens, bp, rng.fill_bytes(32))
    return _ok(verify, be), proof


# ---- BASELINE config 3/4 shape: Edwards scalar_mul of a public base by committed bits
def scalar_mul_roundtrip(be, base, bits, expected, rng=None, cap=None):
    rng = rng or SeededRng(b"smul")
    n = 27 * len(bits) + 4
    cap = cap or (1 << (n - 1).bit_length())
    bp = be.bp_gens(cap)
    prover = be.Prover(be.Transcript(b"ScalarMul"))
    comms, vars_ = [], []
    for b in bits:
        V, var = prover.commit(b, rng.scalar())
        comms.append(V)
        vars_.append(var)
    q = be.EdwardsGadget.scalar_mul(be.EdwardsGadget.from_point(base), vars_, prover)
    q.equal(be.EdwardsGadget.from_point(expected), prover)
    proof = prover.prove(bp, rng.fill_bytes(32))

    def verify():
        v = be.Verifier(be.Transcript(b"ScalarMul"))
        vv = [v.commit(c) for c in comms]
        q2 = be.EdwardsGadget.scalar_mul(be.EdwardsGadget.from_point(base), vv, v)
        q2.equal(be.EdwardsGadget.from_point(expected), v)
        v.verify(proof, be.pc_gens, bp, rng.fill_bytes(32))
    return _ok(verify, be), proof


# ---- randomized (two-phase) constraints.  The reference enables them (Cargo.toml:18 `yoloproofs`,
# ristretto_point.rs:3 imports RandomizedConstraintSystem) and registers none, so it has no test to restate; the case
# below is the k-shuffle the bulletproofs r1cs documentation builds on specify_randomized_constraints:
# prod (x_i - z) == prod (y_i - z) for a challenge z drawn after the first-phase commitments.
def shuffle_gadget(cs, x, y):
    k = len(x)
    assert len(y) == k
    if k == 1:
        cs.constrain(y[0] - x[0])
        return

    def randomized(rcs):
        z = rcs.challenge_scalar(b"shuffle challenge")

        def product(v):
            _, _, out = rcs.multiply(v[k - 1] - z, v[k - 2] - z)
            for i in reversed(range(k - 2)):
                _, _, out = rcs.multiply(out + 0, v[i] - z)
            return out
        rcs.constrain(product(x) - product(y))
    cs.specify_randomized_constraints(randomized)


def shuffle_roundtrip(be, inputs, outputs, first_phase_squares=0, rng=None, cap=64):
    """Proves that `outputs` is a permutation of `inputs` (both committed).  first_phase_squares > 0 also allocates
    that many first-phase multipliers (x_0 * x_0 = committed square), so that a proof carries both phases."""
    rng = rng or SeededRng(b"shuffle")
    bp = be.bp_gens(cap)
    k = len(inputs)
    prover = be.Prover(be.Transcript(b"ShuffleProof"))
    comms, xs, ys = [], [], []
    for v in list(inputs) + list(outputs):
        V, var = prover.commit(v, rng.scalar())
        comms.append(V)
        (xs if len(xs) < k else ys).append(var)
    sq_comms = []
    for _ in range(first_phase_squares):
        V, sq = prover.commit(inputs[0] * inputs[0], rng.scalar())
        sq_comms.append(V)
        _, _, o = prover.multiply(xs[0] + 0, xs[0] + 0)
        prover.constrain(o - sq)
    shuffle_gadget(prover, xs, ys)
    proof = prover.prove(bp, rng.fill_bytes(32))

    def verify():
        v = be.Verifier(be.Transcript(b"ShuffleProof"))
        vv = [v.commit(c) for c in comms]
        for c in sq_comms:
            sq = v.commit(c)
            _, _, o = v.multiply(vv[0] + 0, vv[0] + 0)
            v.constrain(o - sq)
        shuffle_gadget(v, vv[:k], vv[k:])
        v.verify(proof, be.pc_gens, bp, rng.fill_bytes(32))
    return _ok(verify, be), proof


# ---- edge: circuits with no multiplier (n = 0, padded to 1: an inner-product argument of zero rounds) and with one
def linear_only_roundtrip(be, a, b, total, n_mult=0, rng=None, cap=4):
    """commits a, b and proves a + b = total with linear constraints only (n_mult = 0), or also a * b = ab (n_mult = 1)"""
    rng = rng or SeededRng(b"lin")
    bp = be.bp_gens(cap)

    def circuit(cs, va, vb):
        cs.constrain(va + vb - total)
        for _ in range(n_mult):
            _, _, o = cs.multiply(va + 0, vb + 0)
            cs.constrain(o - (a * b))
    prover = be.Prover(be.Transcript(b"Linear"))
    Va, va = prover.commit(a, rng.scalar())
    Vb, vb = prover.commit(b, rng.scalar())
    circuit(prover, va, vb)
    proof = prover.prove(bp, rng.fill_bytes(32))

    def verify():
        v = be.Verifier(be.Transcript(b"Linear"))
        circuit(v, v.commit(Va), v.commit(Vb))
        v.verify(proof, be.pc_gens, bp, rng.fill_bytes(32))
    return _ok(verify, be), proof
