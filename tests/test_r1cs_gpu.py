"""GPU: the R1CS path through the C ABI.

1. The reference's verdict table (tests/edwards_point.rs, scalar.rs, sk_know_gadget.rs)
   holds on the GPU prover + GPU verifier.
2. Proof BYTES equal the oracle's under the same seeded RNG (integer work: byte equality),
   for every reference circuit and for padded / unpadded sizes.
3. Cross-verification: oracle verifies GPU proofs, GPU verifies oracle proofs.
"""
import random

import ctypes

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
    return random.Random(77)


def test_generators_match_oracle(gbe, oracle):
    g = gbe.bp_gens(128)
    G, H = oracle.gens(128)
    B, Bb = oracle.pedersen()
    assert g.export("G", 0, 128) == oracle.encode(G, 128)
    assert g.export("H", 0, 128) == oracle.encode(H, 128)
    assert g.export("B") == oracle.encode(B, 1)
    assert g.export("b") == oracle.encode(Bb, 1)


def test_pedersen_commit_matches_oracle(gbe, obe, rnd):
    gbe.bp_gens(128)
    p = gbe.Prover(b"commit")
    vals = [0, 1, L - 1, 2**252, rnd.randrange(L), rnd.randrange(L)]
    bls = [0, L - 1, 1, rnd.randrange(L), rnd.randrange(L), 0]
    Vs, _ = p.commit_batch(vals, bls)
    for v, b, V in zip(vals, bls, Vs):
        assert V == obe.pc_gens.commit(v, b)
    V1, _ = p.commit(vals[4], bls[4])
    assert V1 == Vs[4]


def _both(case, gbe, obe, *args, seed=b"x", **kw):
    ok_g, pg = case(gbe, *args, rng=R.SeededRng(seed), **kw)
    ok_o, po = case(obe, *args, rng=R.SeededRng(seed), **kw)
    assert ok_g == ok_o
    assert pg.to_bytes() == po.to_bytes()
    return ok_g


def test_conditionally_select(gbe, obe, rnd):
    P1, ID = sonny.random_point(rnd), sonny.EdwardsPoint.identity()
    assert _both(rc.cond_selection_roundtrip, gbe, obe, P1, P1, 1)
    assert not _both(rc.cond_selection_roundtrip, gbe, obe, P1, P1, 0)
    assert _both(rc.cond_selection_roundtrip, gbe, obe, P1, ID, 0)
    assert not _both(rc.cond_selection_roundtrip, gbe, obe, P1, ID, 1)


def test_double_pub_points(gbe, obe, rnd):
    p, bad = sonny.random_point(rnd), sonny.random_point(rnd)
    assert _both(rc.double_roundtrip, gbe, obe, p, p.double())
    assert not _both(rc.double_roundtrip, gbe, obe, p, bad)


def test_add_pub_points(gbe, obe, rnd):
    """BASELINE config 1 shape on the GPU: 16 multipliers (a power of two, no padding)"""
    p1, p2, bad = (sonny.random_point(rnd) for _ in range(3))
    assert _both(rc.addition_roundtrip, gbe, obe, p1, p2, p1 + p2)
    assert not _both(rc.addition_roundtrip, gbe, obe, p1, p2, bad)


def test_point_committing(gbe, obe, rnd):
    A = sonny.random_point(rnd)
    assert _both(rc.point_committing_roundtrip, gbe, obe, A, A)
    assert not _both(rc.point_committing_roundtrip, gbe, obe, A, A + A)


def test_point_curve_eq(gbe, obe, rnd):
    assert _both(rc.curve_eq_roundtrip, gbe, obe, sonny.random_point(rnd))
    assert not _both(rc.curve_eq_roundtrip, gbe, obe, sonny.EdwardsPoint(1, 1, 1, 1))


def test_is_not_zero(gbe, obe, rnd):
    assert _both(rc.is_not_zero_roundtrip, gbe, obe, 1)
    assert _both(rc.is_not_zero_roundtrip, gbe, obe, rnd.randrange(1, L))
    with pytest.raises(RuntimeError):
        rc.is_not_zero_roundtrip(gbe, 0)


def test_ristretto_add_134_multipliers(gbe, obe, rnd):
    """BASELINE config 5 unit proof: 134 multipliers padded to 256"""
    p1, p2 = sonny.RistrettoPoint(sonny.random_point(rnd)), sonny.RistrettoPoint(sonny.random_point(rnd))
    assert _both(rc.ristretto_add_roundtrip, gbe, obe, p1, p2, p1 + p2)
    assert not _both(rc.ristretto_add_roundtrip, gbe, obe, p1, p2, p1 + p1)


def test_scalar_mul_gadget(gbe, obe, rnd):
    """BASELINE config 3 shape, small: committed bits, 27 multipliers per bit"""
    base = sonny.random_point(rnd)
    bits = [1, 0, 1, 1, 0, 1]
    k = sum(b << i for i, b in enumerate(bits))
    assert _both(rc.scalar_mul_roundtrip, gbe, obe, base, bits, base * k)
    assert not _both(rc.scalar_mul_roundtrip, gbe, obe, base, bits, base * (k + 1))
    assert not _both(rc.scalar_mul_roundtrip, gbe, obe, base, [1, 0, 2, 1, 0, 1], base * k)


def test_sk_knowledge_small_vs_oracle(gbe, obe, rnd):
    basep = sonny.RISTRETTO_BASEPOINT
    sk = 0b1011
    assert _both(rc.sk_knowledge_roundtrip, gbe, obe, basep, basep * sk, sk, nbits=4, cap=256)
    assert not _both(rc.sk_knowledge_roundtrip, gbe, obe, basep, basep, sk, nbits=4, cap=256)


def test_sk_knowledge_256_bits(gbe, rnd):
    """ref: sk_know_gadget.rs:110-124 at full size: 256 committed bits, 7506 multipliers, gens 8192"""
    basep = sonny.RISTRETTO_BASEPOINT
    sk = rnd.randrange(1, sonny.R_ORDER)
    pk = basep * sk
    assert rc.sk_knowledge_roundtrip(gbe, basep, pk, sk)[0]
    assert not rc.sk_knowledge_roundtrip(gbe, basep, basep, sk)[0]
    other = sonny.RistrettoPoint(sonny.random_point(rnd))
    assert not rc.sk_knowledge_roundtrip(gbe, basep, other, sk)[0]


def test_cross_verification(gbe, obe, rnd):
    p1, p2 = sonny.random_point(rnd), sonny.random_point(rnd)
    p3 = p1 + p2
    _, proof_g = rc.addition_roundtrip(gbe, p1, p2, p3)
    _, proof_o = rc.addition_roundtrip(obe, p1, p2, p3, rng=R.SeededRng(b"other"))

    def circuit(be, cs):
        g1, g2, g3 = (be.EdwardsGadget.from_point(p) for p in (p1, p2, p3))
        g3.equal(g1.add(g2, cs), cs)
    # oracle verifies the GPU proof
    v = obe.Verifier(obe.Transcript(b"Point Addition"))
    circuit(obe, v)
    assert v.verify(R.R1CSProof.from_bytes(proof_g.to_bytes()), obe.pc_gens, obe.bp_gens(128), bytes(32))
    # GPU verifies the oracle proof
    bp = gbe.bp_gens(128)
    v = gbe.Verifier(gbe.Transcript(b"Point Addition"))
    circuit(gbe, v)
    assert v.verify(proof_o, None, bp, bytes(32))


def test_tampered_proofs_rejected(gbe, rnd):
    p1, p2 = sonny.random_point(rnd), sonny.random_point(rnd)
    p3 = p1 + p2
    ok, proof = rc.addition_roundtrip(gbe, p1, p2, p3)
    assert ok
    raw = proof.to_bytes()
    bp = gbe.bp_gens(128)
    from bulletproofs_gadgets_amd import BpgError

    def verify(b):
        v = gbe.Verifier(gbe.Transcript(b"Point Addition"))
        g1, g2, g3 = (gbe.EdwardsGadget.from_point(p) for p in (p1, p2, p3))
        g3.equal(g1.add(g2, v), v)
        try:
            return v.verify(gbe.g.Proof(b), None, bp, bytes(32))
        except (gbe.VerificationError, BpgError):
            return False
    assert verify(raw)
    for off in range(1, len(raw), 32):
        bad = bytearray(raw)
        bad[off] ^= 1
        assert not verify(bytes(bad)), off
    assert not verify(raw[:-32])
    assert not verify(b"\x02" + raw[1:])


def test_generators_too_short(gbe, rnd):
    from bulletproofs_gadgets_amd import BpgError, _native
    p = sonny.random_point(rnd)
    with pytest.raises(BpgError) as e:
        rc.double_roundtrip(gbe, p, p.double(), cap=8)
    assert e.value.code == _native.EGENS


def test_concurrent_provers_share_generators(gbe, obe, rnd):
    """N host threads, each driving its own proof on its own stream against one
    shared immutable generator object; every thread's proof bytes equal the serial run's."""
    import threading
    gbe.bp_gens(256)
    pts = [(sonny.random_point(rnd), sonny.random_point(rnd)) for _ in range(6)]
    rpts = [(sonny.RistrettoPoint(a), sonny.RistrettoPoint(b)) for a, b in pts]

    def work(i):
        if i % 2:
            a, b = rpts[i]
            return rc.ristretto_add_roundtrip(gbe, a, b, a + b, rng=R.SeededRng(b"t%d" % i), cap=256)
        a, b = pts[i]
        return rc.addition_roundtrip(gbe, a, b, a + b, rng=R.SeededRng(b"t%d" % i), cap=256)
    serial = [work(i) for i in range(6)]
    for _ in range(2):
        res = [None] * 6
        errs = []

        def run(i):
            try:
                res[i] = work(i)
            except Exception as e:      # noqa: BLE001
                errs.append(e)
        th = [threading.Thread(target=run, args=(i,)) for i in range(6)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, errs
        for i in range(6):
            assert res[i][0] and serial[i][0]
            assert res[i][1].to_bytes() == serial[i][1].to_bytes()


def _radd_circuit(be, cs, p1, p2, p3):
    g1, g2, g3 = (be.RistrettoGadget.from_point(p, cs) for p in (p1, p2, p3))
    g3.equals(cs, g1.add(cs, g2))


def _eadd_circuit(be, cs, p1, p2, p3):
    g1, g2, g3 = (be.EdwardsGadget.from_point(p) for p in (p1, p2, p3))
    g3.equal(g1.add(g2, cs), cs)


def test_verify_batch(gbe, obe, rnd):
    """BASELINE config 5: k proofs (mixed circuits and sizes: 134 -> 256 and 16 multipliers, with and
    without commitments) checked by ONE combined multiscalar multiplication.  The batch verdict must be
    the AND of the per-proof verdicts (which the tests above pin against the oracle)."""
    from bulletproofs_gadgets_amd import r1cs as g
    bp = gbe.bp_gens(256)
    items = []          # (build verifier, proof)
    for i in range(5):
        a, b = sonny.random_point(rnd), sonny.random_point(rnd)
        if i % 2 == 0:
            ra, rb = sonny.RistrettoPoint(a), sonny.RistrettoPoint(b)
            ok, proof = rc.ristretto_add_roundtrip(gbe, ra, rb, ra + rb, rng=R.SeededRng(b"vb%d" % i), cap=256)
            items.append((lambda ra=ra, rb=rb: (b"RistrettoAdd", lambda v: _radd_circuit(gbe, v, ra, rb, ra + rb)), proof))
        else:
            ok, proof = rc.addition_roundtrip(gbe, a, b, a + b, rng=R.SeededRng(b"vb%d" % i), cap=256)
            items.append((lambda a=a, b=b: (b"Point Addition", lambda v: _eadd_circuit(gbe, v, a, b, a + b)), proof))
        assert ok
    # one proof with commitments (m = 8)
    A = sonny.random_point(rnd)
    p2 = gbe.Prover(b"PointDouble")
    srng = R.SeededRng(b"vbc")
    g1, c1 = gbe.EdwardsGadget.prover_commit_to_sonny_edwards_point(p2, A, srng)
    g2, c2 = gbe.EdwardsGadget.prover_commit_to_sonny_edwards_point(p2, A, srng)
    g1.equal(g2, p2)
    cproof = p2.prove(bp, bytes(32))

    def commit_circuit(v):
        h1 = gbe.EdwardsGadget.verifier_commit_to_sonny_edwards_point(v, c1)
        h2 = gbe.EdwardsGadget.verifier_commit_to_sonny_edwards_point(v, c2)
        h1.equal(h2, v)
    items.append((lambda: (b"PointDouble", commit_circuit), cproof))

    def verifiers():
        out = []
        for mk, _ in items:
            label, circuit = mk()
            v = gbe.Verifier(label)
            circuit(v)
            out.append(v)
        return out
    proofs = [p for _, p in items]
    ext = bytes(range(1, 65)) * len(items)
    assert g.verify_batch(verifiers(), proofs, bp, ext)
    assert g.verify_batch([], [], bp)
    assert g.verify_batch(verifiers()[:1], proofs[:1], bp, ext)
    # any single bad proof fails the whole batch, wherever it sits
    for pos in (0, 3, len(items) - 1):
        raw = bytearray(proofs[pos].to_bytes())
        raw[1 + 32 * 8 + 5] ^= 4            # t_x
        bad = list(proofs)
        bad[pos] = g.Proof(bytes(raw))
        with pytest.raises(g.VerificationError):
            g.verify_batch(verifiers(), bad, bp, ext)
    # proofs swapped between two statements of the same shape
    sw = list(proofs)
    sw[0], sw[2] = sw[2], sw[0]
    with pytest.raises(g.VerificationError):
        g.verify_batch(verifiers(), sw, bp, ext)
    # a point that does not decode
    raw = bytearray(proofs[1].to_bytes())
    raw[1:33] = b"\xff" * 32
    bad = list(proofs)
    bad[1] = g.Proof(bytes(raw))
    with pytest.raises(g.VerificationError):
        g.verify_batch(verifiers(), bad, bp, ext)
    # default arguments, and an all-zero external stream, must still check EVERY proof -- the
    # batching weights come from each proof's transcript RNG, not from the caller's bytes (which were once taken
    # as the weights themselves: zero bytes gave rho_j = 0 and proofs at index >= 1 went unchecked)
    assert g.verify_batch(verifiers(), proofs, bp)
    for pos in (1, 2, len(items) - 1):
        raw = bytearray(proofs[pos].to_bytes())
        raw[1 + 32 * 9 + 3] ^= 1            # t_x_blinding
        bad = list(proofs)
        bad[pos] = g.Proof(bytes(raw))
        with pytest.raises(g.VerificationError):
            g.verify_batch(verifiers(), bad, bp)
        with pytest.raises(g.VerificationError):
            g.verify_batch(verifiers(), bad, bp, bytes(32 * len(items)))
    # a stream that runs out is an error, never zero padding
    with pytest.raises(ValueError):
        g.verify_batch(verifiers(), proofs, bp, ext[:32 * len(items) - 1])
    # no device allocation on the hot verify path -- after the calls above have sized the lane's
    # buffers, 100 more calls leave the free device memory where it was
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    free0, free1, total = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
    gbe.ctx.synchronize()
    assert hip.hipMemGetInfo(ctypes.byref(free0), ctypes.byref(total)) == 0
    for _ in range(100):
        assert g.verify_batch(verifiers(), proofs, bp, ext)
    assert hip.hipMemGetInfo(ctypes.byref(free1), ctypes.byref(total)) == 0
    assert free1.value == free0.value


def test_prove_many_equals_sequential_proves(gbe, rnd):
    """bpg_prover_prove_many: k proofs by one thread with their blinding RNGs drawn in lock-step (eight Keccak
    states per permutation).  Bytes must equal those of k separate prove() calls fed the same external randomness,
    which the tests above pin against the oracle: groups of fewer than, exactly and more than eight provers of
    one circuit (different witnesses), provers with committed variables (the RNG is rekeyed with their blindings),
    and mixed circuit shapes (those fall back to one RNG at a time)."""
    import hashlib
    from bulletproofs_gadgets_amd import r1cs as g
    bp = gbe.bp_gens(256)

    def build(kind, a, b, seed):
        if kind == "radd":
            p = gbe.Prover(b"RistrettoAdd")
            ra, rb = sonny.RistrettoPoint(a), sonny.RistrettoPoint(b)
            _radd_circuit(gbe, p, ra, rb, ra + rb)
        elif kind == "eadd":
            p = gbe.Prover(b"Point Addition")
            _eadd_circuit(gbe, p, a, b, a + b)
        else:
            p = gbe.Prover(b"PointDouble")
            srng = R.SeededRng(seed)
            g1, _ = gbe.EdwardsGadget.prover_commit_to_sonny_edwards_point(p, a, srng)
            g2, _ = gbe.EdwardsGadget.prover_commit_to_sonny_edwards_point(p, a, srng)
            g1.equal(g2, p)
        return p

    for kinds in (["radd"], ["radd"] * 3, ["eadd"] * 8, ["radd"] * 11, ["commit"] * 5, ["radd", "eadd", "commit", "radd", "eadd"],
                  ["eadd"] * 8 + ["radd"] * 2):
        k = len(kinds)
        wit = [(sonny.random_point(rnd), sonny.random_point(rnd)) for _ in range(k)]
        ext = b"".join(hashlib.sha256(b"pm%d-%d" % (k, j)).digest() for j in range(k))
        one_by_one = [build(kinds[j], *wit[j], b"s%d" % j).prove(bp, ext[32 * j:32 * j + 32]).to_bytes() for j in range(k)]
        many = g.Prover.prove_many([build(kinds[j], *wit[j], b"s%d" % j) for j in range(k)], ext)
        assert [p.to_bytes() for p in many] == one_by_one, kinds
    assert g.Prover.prove_many([]) == []


@pytest.mark.parametrize("fold_round", [1, 2, 3])
def test_materialized_folded_generators(oracle, obe, fold_round):
    """The prover materializes the folded generators G'_i = sum_{j = i mod m} s_j G_j after an inner-product round
    (by default only from 2^12 multipliers on; forced here through BPG_IPP_FOLD so that oracle-sized circuits take
    that path): proof bytes must still equal the oracle's, for single proofs, for groups proved by prove_many (the
    later rounds then run one MSM over the union of the group's folded tables), with and without commitments."""
    import hashlib
    import os
    import random
    import bulletproofs_gadgets_amd as bpg
    from bulletproofs_gadgets_amd import r1cs as g
    old = os.environ.get("BPG_IPP_FOLD")
    os.environ["BPG_IPP_FOLD"] = str(fold_round)
    try:
        ctx = bpg.Context(0)              # read when a context is created
    finally:
        if old is None:
            os.environ.pop("BPG_IPP_FOLD", None)
        else:
            os.environ["BPG_IPP_FOLD"] = old
    try:
        be = GpuBackend(ctx)
        rnd = random.Random(31 + fold_round)
        a, b = sonny.random_point(rnd), sonny.random_point(rnd)
        ra, rb = sonny.RistrettoPoint(a), sonny.RistrettoPoint(b)
        bits = [1, 0, 1, 1, 0, 1, 1]
        k = sum(x << i for i, x in enumerate(bits))
        for case, args in ((rc.ristretto_add_roundtrip, (ra, rb, ra + rb)), (rc.scalar_mul_roundtrip, (a, bits, a * k))):
            ok_g, pg = case(be, *args, rng=R.SeededRng(b"fold"), cap=256)
            ok_o, po = case(obe, *args, rng=R.SeededRng(b"fold"), cap=256)
            assert ok_g and ok_o and pg.to_bytes() == po.to_bytes()
        # groups: the same statements proved one by one on the default context's path are pinned above; here the
        # grouped, folded path must give the bytes of the single, folded path
        bp = be.bp_gens(256)

        def build(x, y):
            p = be.Prover(b"RistrettoAdd")
            _radd_circuit(be, p, sonny.RistrettoPoint(x), sonny.RistrettoPoint(y), sonny.RistrettoPoint(x) + sonny.RistrettoPoint(y))
            return p
        wit = [(sonny.random_point(rnd), sonny.random_point(rnd)) for _ in range(5)]
        ext = b"".join(hashlib.sha256(b"fg%d" % j).digest() for j in range(5))
        single = [build(*wit[j]).prove(bp, ext[32 * j:32 * j + 32]).to_bytes() for j in range(5)]
        assert [p.to_bytes() for p in g.Prover.prove_many([build(*w) for w in wit], ext)] == single
    finally:
        ctx.close()


def test_allocation_failure_releases_jobs_and_workspaces(gpu_ctx, rnd):
    """VERDICT r1 #7 / ADVICE r1: a device allocation that fails in the middle of bpg_msm_submit (or of a prove /
    verify that calls it) must hand the job back to its lane: the same call succeeds afterwards, with the same
    bytes, and the failure is BPG_ENOMEM / BPG_EDEVICE, not a hang or a leaked slot."""
    import bulletproofs_gadgets_amd as bpg
    from bulletproofs_gadgets_amd import _native, synth
    lib = _native.lib()
    ctx = bpg.Context(0)          # a fresh context: its lanes and MSM jobs have no workspace yet, everything has to grow
    try:
        n = 3000
        pts = ctx.points_from_uniform(synth.uniform_bytes(n, tag=b"af"))
        sc = synth.scalars(n, tag=b"af-s")
        want = gpu_ctx.msm(gpu_ctx.points_from_uniform(synth.uniform_bytes(n, tag=b"af")), sc)
        # countdown 2: the first workspace still to grow grows, the next allocation fails -- repeated, this walks
        # through every allocation site of the submit path, one per attempt, until nothing is left to grow
        failures = 0
        for _ in range(24):
            lib.bpg_debug_fail_alloc(2 if failures else 1)
            try:
                got = ctx.msm(pts, sc)
                break
            except bpg.BpgError as e:
                assert e.code in (_native.EDEVICE, _native.ENOMEM)
                failures += 1
            finally:
                lib.bpg_debug_fail_alloc(0)
        assert failures >= 4 and got == want
        # a larger MSM afterwards: the buffers grow again, through the same failures, and the small one still works between them
        n2 = 40000
        pts2 = ctx.points_from_uniform(synth.uniform_bytes(n2, tag=b"af2"))
        sc2 = synth.scalars(n2, tag=b"af2-s")
        failures = 0
        for _ in range(24):
            lib.bpg_debug_fail_alloc(2 if failures else 1)
            try:
                big = ctx.msm(pts2, sc2)
                break
            except bpg.BpgError as e:
                assert e.code in (_native.EDEVICE, _native.ENOMEM)
                failures += 1
            finally:
                lib.bpg_debug_fail_alloc(0)
            assert ctx.msm(pts, sc) == want          # the lane's job pool still works after every failure
        assert failures >= 3
        assert big == ctx.msm(pts2, sc2) == gpu_ctx.msm(gpu_ctx.points_from_uniform(synth.uniform_bytes(n2, tag=b"af2")), sc2)
    finally:
        ctx.close()
    # prove / verify: the failure surfaces as an error code and the next proof is byte-identical to a clean one
    gbe = GpuBackend(gpu_ctx)
    a, b = sonny.random_point(rnd), sonny.random_point(rnd)
    ok, clean = rc.addition_roundtrip(gbe, a, b, a + b, rng=R.SeededRng(b"af"), cap=128)
    assert ok
    for k in (1, 2, 3):
        ctx2 = bpg.Context(0)                        # fresh lanes: their workspaces have to be allocated
        try:
            be2 = GpuBackend(ctx2)
            be2.bp_gens(128)
            lib.bpg_debug_fail_alloc(k)
            try:
                with pytest.raises(bpg.BpgError):
                    rc.addition_roundtrip(be2, a, b, a + b, rng=R.SeededRng(b"af"), cap=128)
            finally:
                lib.bpg_debug_fail_alloc(0)
            ok2, again = rc.addition_roundtrip(be2, a, b, a + b, rng=R.SeededRng(b"af"), cap=128)
            assert ok2 and again.to_bytes() == clean.to_bytes()
        finally:
            ctx2.close()


def test_circuit_structure_cache(gpu_ctx, gbe, obe, rnd):
    """a second proof of the same circuit reuses the resident term lists, whatever its public constants
    and witness are; the bytes stay those of the oracle.  A different circuit is a different entry."""
    h0, m0 = gpu_ctx.flat_cache_stats()
    p1, p2, q1, q2 = (sonny.random_point(rnd) for _ in range(4))
    assert _both(rc.addition_roundtrip, gbe, obe, p1, p2, p1 + p2, seed=b"c1")
    h1, m1 = gpu_ctx.flat_cache_stats()
    assert _both(rc.addition_roundtrip, gbe, obe, q1, q2, q1 + q2, seed=b"c2")     # other points, same circuit
    h2, m2 = gpu_ctx.flat_cache_stats()
    assert m2 == m1 and h2 >= h1 + 2                     # prover and verifier both found it
    assert m1 - m0 <= 2                                  # at most: the prover's plan and the verifier's (with constants)
    assert _both(rc.double_roundtrip, gbe, obe, p1, p1.double(), seed=b"c3")
    h3, m3 = gpu_ctx.flat_cache_stats()
    assert m3 >= m2 + 1 or h3 > h2                       # another circuit: its own entry (or already resident from an earlier test)
    # more circuits than the cache holds: the oldest entries go, results stay right
    for k in range(2, 13):
        bits = [rnd.randrange(2) for _ in range(k)]
        base = sonny.random_point(rnd)
        assert _both(rc.scalar_mul_roundtrip, gbe, obe, base, bits, base * sum(b << i for i, b in enumerate(bits)), seed=b"c%d" % k)
    assert _both(rc.addition_roundtrip, gbe, obe, p1, p2, p1 + p2, seed=b"c1")


def test_prove_names_the_generators(gbe, obe, rnd):
    """Prover::new takes only the Pedersen bases, Prover::prove(&bp_gens) the generator vectors (ref:
    tests/edwards_point.rs:28 vs :47): a prover made against the capacity-1 object proves against another one."""
    g = gbe.g
    small = g.BulletproofGens(gbe.ctx, 1, 1)
    big = gbe.bp_gens(128)
    p1, p2 = sonny.random_point(rnd), sonny.random_point(rnd)
    rng = R.SeededRng(b"add")
    prover = g.Prover(small, b"Point Addition")
    g1, g2, g3 = (g.SonnyEdwardsPointGadget.from_point(p) for p in (p1, p2, p1 + p2))
    g3.equal(g1.add(g2, prover), prover)
    with pytest.raises(RuntimeError, match="BPG_EGENS"):
        q = g.Prover(small, b"Point Addition")
        h1, h2, h3 = (g.SonnyEdwardsPointGadget.from_point(p) for p in (p1, p2, p1 + p2))
        h3.equal(h1.add(h2, q), q)
        q.prove(small, bytes(32))                                   # 16 multipliers do not fit capacity 1
    proof = prover.prove(big, rng.fill_bytes(32))
    ok, want = rc.addition_roundtrip(obe, p1, p2, p1 + p2, rng=R.SeededRng(b"add"))
    assert ok and proof.to_bytes() == want.to_bytes()


def test_missing_assignment(gbe):
    """R1CSError::MissingAssignment: a prover asked for a multiplier without its inputs (ConstraintSystem::
    allocate_multiplier(None)); the reference reaches it through from_lcs on a prover, whose nonzero gadget has no
    assignment to give (ref: ristretto_point.rs:34-45 -> :50-90 -> scalar.rs:9-26)."""
    from bulletproofs_gadgets_amd import BpgError, _native
    gbe.bp_gens(64)
    p = gbe.Prover(b"missing")
    with pytest.raises(BpgError) as e:
        p.allocate_multiplier(None)
    assert e.value.code == _native.EMISSING
    gp = gbe.g.SonnyEdwardsPointGadget.from_point(sonny.BASEPOINT)
    with pytest.raises(BpgError) as e:
        gbe.g.SonnyRistrettoPointGadget.from_lcs(gp, 4, p)
    assert e.value.code in (_native.EMISSING, _native.EGADGET)
    # the same calls on a verifier allocate unassigned multipliers
    v = gbe.Verifier(b"missing")
    v.allocate_multiplier(None)
    gbe.g.SonnyRistrettoPointGadget.from_lcs(gp, 4, v)


def test_circuits_without_multipliers(gbe, obe):
    """edge sizes: no multiplier at all (n = 0 -> padded to 1, zero inner-product rounds), one, three (padded to 4)"""
    for n_mult in (0, 1, 3):
        assert _both(rc.linear_only_roundtrip, gbe, obe, 5, 7, 12, n_mult=n_mult)
        assert not _both(rc.linear_only_roundtrip, gbe, obe, 5, 7, 13, n_mult=n_mult)


def test_early_blinding_draws_give_the_same_proof(gpu_ctx):
    """bpg_prover_begin_blindings: the helper thread's draws, streamed to the device during prove(), give the bytes of a
    plain prove() with the same external randomness -- also when the helper is stopped by a later commit, when the
    call comes before any gadget call or after all of them, and when the prover is handed to prove_many."""
    from bulletproofs_gadgets_amd import r1cs as g
    from tools import sharded_proof as sp
    nbits = 40                                     # 1084 multipliers: above the raw-draw threshold
    stmt = sp.statement(nbits)
    base, expected, vals, bl, ext_p, ext_v = stmt
    E = g.SonnyEdwardsPointGadget
    gens = g.BulletproofGens(gpu_ctx, 2048, 1)
    try:
        want, *_ = sp.prove_and_verify(g, gens, nbits, stmt, early=False)
        got, *_ = sp.prove_and_verify(g, gens, nbits, stmt, early=True)
        assert got.to_bytes() == want.to_bytes()

        def build(when):
            p = g.Prover(gens, b"ScalarMul")
            half = len(vals) // 64 * 32
            if when == "before_last_commit":       # a commit after the call: the helper is stopped, prove() draws itself
                _, v1 = p.commit_batch_raw(vals[:half], bl[:half])
                p.begin_blindings(ext_p)
                _, v2 = p.commit_batch_raw(vals[half:], bl[half:])
                both = list(v1)[:half // 32] + list(v2)[:nbits - half // 32]
                vars_ = (ctypes.c_uint64 * nbits)(*both)
            else:
                _, vars_ = p.commit_batch_raw(vals, bl)
            if when == "before_gadget":
                p.begin_blindings(ext_p)
            E.scalar_mul(E.from_point(base), vars_, p).equal(E.from_point(expected), p)
            if when == "after_gadget":
                p.begin_blindings(ext_p)
            return p
        for when in ("before_gadget", "after_gadget", "before_last_commit"):
            assert build(when).prove(gens, b"\x55" * 32).to_bytes() == want.to_bytes(), when       # prove's own rng is not asked
        # prove_many stops the helpers and draws in lock-step, from the bytes handed over early
        ps = [build("before_gadget"), build("before_gadget")]
        proofs = g.Prover.prove_many(ps, b"\x55" * 64)
        assert [q.to_bytes() for q in proofs] == [want.to_bytes()] * 2
        with pytest.raises(Exception):
            p = build("before_gadget")
            p.begin_blindings(ext_p)               # twice
        del p                                      # a prover dropped with its helper running
    finally:
        gens.free()


def test_oversized_generator_table_fails_at_once(gpu_ctx):
    """a request the device cannot hold is refused without asking the driver (device_malloc checks the free memory):
    capacity 2^26 would need 36 blocks x 2^27 records x 128 bytes = 618 GB"""
    import time
    import bulletproofs_gadgets_amd as bpg
    from bulletproofs_gadgets_amd import r1cs as g
    t0 = time.perf_counter()
    with pytest.raises(bpg.BpgError) as e:
        g.BulletproofGens(gpu_ctx, 1 << 26, 1)
    assert "ENOMEM" in str(e.value)
    assert time.perf_counter() - t0 < 5.0
    gens = g.BulletproofGens(gpu_ctx, 64, 1)          # the context is as usable as before
    gens.free()


@pytest.mark.parametrize("nbits,k", [(40, 19), (150, 17)])
def test_prove_many_in_chunks_of_sixteen(gpu_ctx, nbits, k):
    """prove_many over more provers than one chunk holds (16), at sizes where the raw blinding draws go through the
    lane's pinned staging (>= 256 multipliers) and, at 150 bits, where the generators are folded (half-window table):
    chunks of 16, RNG lanes of 8 and a remainder must give the bytes of k separate prove() calls."""
    import hashlib
    from bulletproofs_gadgets_amd import r1cs as g
    from tools import sharded_proof as sp
    base, expected, vals, _, _, _ = sp.statement(nbits)
    E = g.SonnyEdwardsPointGadget
    n = 27 * nbits + 4
    gens = g.BulletproofGens(gpu_ctx, 1 << (n - 1).bit_length(), 1)
    try:
        def build(j):
            raw = hashlib.shake_256(b"chunks%d/%d" % (nbits, j)).digest(31 * nbits)
            bl = b"".join(raw[31 * i:31 * i + 31] + b"\0" for i in range(nbits))
            p = g.Prover(gens, b"ScalarMul")
            _, vars_ = p.commit_batch_raw(vals, bl)
            E.scalar_mul(E.from_point(base), vars_, p).equal(E.from_point(expected), p)
            return p
        ext = b"".join(hashlib.sha256(b"ext%d" % j).digest() for j in range(k))
        single = [build(j).prove(gens, ext[32 * j:32 * j + 32]).to_bytes() for j in range(k)]
        many = g.Prover.prove_many([build(j) for j in range(k)], ext)
        assert [p.to_bytes() for p in many] == single
        assert len(set(single)) == k
    finally:
        gens.free()


def test_many_threads_groups_early_draws_and_batch_verification(gpu_ctx):
    """Twelve host threads at once, each proving a group of four 4054-multiplier statements (folded generators, pinned
    staging, batched small transfers), every third thread also proving one statement alone with early blinding draws
    (the context's pool of draw buffers), each thread checking its proofs with one verify_batch: bytes equal the
    serial run's, every batch accepted."""
    import hashlib
    import threading
    from bulletproofs_gadgets_amd import r1cs as g
    from tools import sharded_proof as sp
    nbits, T, K = 150, 12, 4
    base, expected, vals, _, _, _ = sp.statement(nbits)
    E = g.SonnyEdwardsPointGadget
    gens = g.BulletproofGens(gpu_ctx, 4096, 1)

    def build(tag, early_ext=None):
        raw = hashlib.shake_256(b"mt/" + tag).digest(31 * nbits)
        bl = b"".join(raw[31 * i:31 * i + 31] + b"\0" for i in range(nbits))
        p = g.Prover(gens, b"ScalarMul")
        comms, vars_ = p.commit_batch_raw(vals, bl)
        if early_ext is not None:
            p.begin_blindings(early_ext)
        E.scalar_mul(E.from_point(base), vars_, p).equal(E.from_point(expected), p)
        return p, comms

    def work(t):
        ext = b"".join(hashlib.sha256(b"mt-ext%d/%d" % (t, j)).digest() for j in range(K))
        built = [build(b"%d/%d" % (t, j)) for j in range(K)]
        proofs = g.Prover.prove_many([p for p, _ in built], ext)
        out = [q.to_bytes() for q in proofs]
        if t % 3 == 0:
            p, _ = build(b"%d/0" % t, early_ext=ext[:32])
            assert p.prove(gens, bytes(32)).to_bytes() == out[0]
        vs = []
        for _, comms in built:
            v = g.Verifier(b"ScalarMul")
            E.scalar_mul(E.from_point(base), v.commit_batch_raw(comms.raw), v).equal(E.from_point(expected), v)
            vs.append(v)
        assert g.verify_batch(vs, proofs, gens, hashlib.shake_256(b"w%d" % t).digest(64 * K))
        return out
    try:
        serial = [work(t) for t in range(3)]
        res, errs = [None] * T, []

        def run(t):
            try:
                res[t] = work(t)
            except Exception as e:      # noqa: BLE001
                import traceback
                traceback.print_exc()
                errs.append(repr(e))
        th = [threading.Thread(target=run, args=(t,)) for t in range(T)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, errs
        assert res[:3] == serial
        assert len({b for r in res for b in r}) == T * K
    finally:
        gens.free()
