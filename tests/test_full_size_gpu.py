"""GPU: BASELINE configs 3, 4 and 5 at their FULL sizes, as tests (the full-size byte checks lived only in
bench.py's output).

  config 3  one proof of the Edwards scalar_mul gadget at 2427 committed bits = 65 533 multipliers (n+ = 2^16):
            proof bytes equal the oracle's under the same seeded randomness (integer work: byte equality), the
            verifier accepts it, and -- the shape of the reference's only large test,
            src/gadgets/sk_knowledge/sk_know_gadget.rs:110-124 -- rejects the statement with the base point or a
            random point in place of the product.
  config 4  one proof at 38 836 bits = 1 048 576 multipliers (n+ = 2^20) with the generators split over 2 ranks
            (threads standing in for processes; the all-gather is a barrier): every rank's bytes equal the
            unsharded prover's.
  config 5  4096 Ristretto-gadget proofs verified by one combined multiscalar multiplication: accepted, and
            rejected when one proof in the middle is corrupted.
"""
import hashlib
import threading
import time

import pytest

from tests.test_sharded_proof_gpu import ThreadExchange

pytestmark = pytest.mark.gpu


def _scalar_mul(g, cs, base, bit_vars, expected):
    E = g.SonnyEdwardsPointGadget
    E.scalar_mul(E.from_point(base), bit_vars, cs).equal(E.from_point(expected), cs)


def test_config3_full_size_proof_bytes_equal_oracle(gpu_ctx, oracle):
    import bench
    from bulletproofs_gadgets_amd import r1cs as g
    from oracle import sonny
    nbits = bench.PROOF_BITS
    base, bits, expected, vals, randomness = bench.proof_statement_bytes(nbits)
    bl, ext_p, ext_v = randomness(b"parity")
    n = 27 * nbits + 4
    assert n == 65533
    gens = g.BulletproofGens(gpu_ctx, 1 << 16, 1)
    try:
        def prove(expect):
            p = g.Prover(gens, b"ScalarMul")
            comms, vars_ = p.commit_batch_raw(vals, bl)
            _scalar_mul(g, p, base, vars_, expect)
            assert p.stats()[0] == n
            return comms, p.prove(gens, ext_p)

        def verify(comms, proof, expect):
            v = g.Verifier(b"ScalarMul")
            _scalar_mul(g, v, base, v.commit_batch_raw(comms.raw), expect)
            return v.verify(proof, None, gens, ext_v)
        comms, proof = prove(expected)
        assert verify(comms, proof, expected)
        t0 = time.perf_counter()
        cpu_s, want = bench.proof_cpu_baseline(nbits)          # the oracle's prover on the same statement and randomness
        assert proof.to_bytes() == want, "GPU proof bytes differ from the oracle's at 2^16 multipliers"
        print("oracle prove: %.1f s" % (time.perf_counter() - t0))
        # negative shape of sk_know_gadget.rs:110-124: the right proof against a wrong public point, and proofs
        # made for a wrong point (prove() succeeds on an unsatisfied circuit; every failure shows in verify)
        import random
        wrong = [base, sonny.random_point(random.Random(5))]
        for w in wrong:
            with pytest.raises(g.VerificationError):
                verify(comms, proof, w)
            c2, p2 = prove(w)
            with pytest.raises(g.VerificationError):
                verify(c2, p2, w)
    finally:
        gens.free()


def test_config4_full_size_sharded_proof_equals_unsharded(gpu_ctx):
    from bulletproofs_gadgets_amd import r1cs as g
    from tools import sharded_proof as sp
    nbits = 38836
    n = 27 * nbits + 4
    assert n == 1 << 20
    stmt = sp.statement(nbits)
    full = g.BulletproofGens(gpu_ctx, n, 1)
    try:
        ref, *_ = sp.prove_and_verify(g, full, nbits, stmt)
    finally:
        full.free()
    world = 2
    ex = ThreadExchange(world)
    res, errs = [None] * world, []

    def work(r):
        try:
            gens = g.BulletproofGens(gpu_ctx, n, 1, shard=(r, world, ex.all_gather(r)))
            try:
                proof, *_ = sp.prove_and_verify(g, gens, nbits, stmt)       # verifies on every rank too
                res[r] = proof.to_bytes()
            finally:
                gens.free()
        except Exception as e:      # noqa: BLE001
            import traceback
            traceback.print_exc()
            errs.append(e)
            ex.bar.abort()
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    assert res[0] == res[1] == ref.to_bytes()
    print("2^20-multiplier proof sha256", hashlib.sha256(res[0]).hexdigest())


def test_config5_full_size_batch_of_4096(gpu_ctx):
    from bulletproofs_gadgets_amd import r1cs as g
    from tools import batch_verify as bv
    gens = g.BulletproofGens(gpu_ctx, 256, 1)
    try:
        r = bv.run(gpu_ctx, gens, 4096, 16)
    finally:
        gens.free()
    assert r["proofs"] == 4096 and r["batch_ok"] and r["bad_batch_rejected"], r
