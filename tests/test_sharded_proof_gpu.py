"""GPU: ONE proof split by generator range over `world` ranks (BASELINE config 4).

Here the ranks are host threads of one process sharing cuda:0, each with its own sharded
generator object (its own range of G / H resident) and its own prover; the partial sums are
exchanged through a thread barrier standing in for the RCCL all-gather.  Every rank must
produce the proof bytes of the unsharded GPU prover, which test_r1cs_gpu.py pins against the
oracle -- and here again directly.  tests/test_sharded_procs_gpu.py runs the same thing as
separate processes over torch.distributed.
"""
import threading

import pytest

from oracle import r1cs as R
from oracle import sonny
from tests import ref_cases as rc
from tests.backends import GpuBackend, OracleBackend

pytestmark = pytest.mark.gpu


class ThreadExchange:
    def __init__(self, world):
        self.world, self.slots, self.bar = world, [None] * world, threading.Barrier(world)

    def all_gather(self, rank):
        def fn(mine):
            self.slots[rank] = mine
            self.bar.wait(timeout=60)
            res = b"".join(self.slots)
            self.bar.wait(timeout=60)
            return res
        return fn


class ShardedBackend(GpuBackend):
    """GpuBackend whose generator objects hold rank's range only"""

    def __init__(self, ctx, rank, world, ex):
        super().__init__(ctx)
        self.rank, self.world, self.ex = rank, world, ex

    def bp_gens(self, cap):
        if cap not in self._gens:
            self._gens[cap] = self.g.BulletproofGens(self.ctx, cap, 1, shard=(self.rank, self.world, self.ex.all_gather(self.rank)))
        self._cur = self._gens[cap]
        return self._cur


def run_ranks(gpu_ctx, world, fn):
    ex = ThreadExchange(world)
    res, errs = [None] * world, []

    def work(r):
        try:
            res[r] = fn(ShardedBackend(gpu_ctx, r, world, ex))
        except Exception as e:      # noqa: BLE001
            import traceback
            traceback.print_exc()
            errs.append(e)
            ex.bar.abort()
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    return res


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_generators_are_slices(gpu_ctx, oracle, world):
    cap = 96          # not a power of two, not divisible by 64
    G, H = oracle.gens(cap)
    encG, encH = oracle.encode(G, cap), oracle.encode(H, cap)

    def fn(be):
        g = be.bp_gens(cap)
        lo, hi = cap * be.rank // world, cap * (be.rank + 1) // world
        assert g.export("G", lo, hi - lo) == encG[32 * lo:32 * hi]
        assert g.export("H", lo, hi - lo) == encH[32 * lo:32 * hi]
        from bulletproofs_gadgets_amd import BpgError
        if lo > 0:
            with pytest.raises(BpgError):
                g.export("G", lo - 1, 1)
        return True
    assert all(run_ranks(gpu_ctx, world, fn))


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_proof_bytes_equal_unsharded(gpu_ctx, oracle, world):
    import random
    rnd = random.Random(5)
    p1, p2 = sonny.random_point(rnd), sonny.random_point(rnd)
    base = sonny.random_point(rnd)
    bits = [1, 0, 1, 1, 0, 1, 1]
    k = sum(b << i for i, b in enumerate(bits))
    obe = OracleBackend(oracle)

    def cases(be):
        out = []
        # 16 multipliers, cap 128: with 4 ranks only rank 0 holds generators the proof uses
        out.append(rc.addition_roundtrip(be, p1, p2, p1 + p2, rng=R.SeededRng(b"sh1")))
        out.append(rc.addition_roundtrip(be, p1, p2, p1 + p1, rng=R.SeededRng(b"sh2")))        # must fail on every rank
        # 193 multipliers padded to 256 = cap, 7 commitments
        out.append(rc.scalar_mul_roundtrip(be, base, bits, base * k, rng=R.SeededRng(b"sh3")))
        out.append(rc.scalar_mul_roundtrip(be, base, bits, base * (k + 1), rng=R.SeededRng(b"sh4")))
        return [(ok, proof.to_bytes()) for ok, proof in out]
    want = cases(obe)
    assert [ok for ok, _ in want] == [True, False, True, False]
    assert cases(GpuBackend(gpu_ctx)) == want
    for got in run_ranks(gpu_ctx, world, cases):
        assert got == want


def test_sharded_batch_verify(gpu_ctx):
    """the combined MSM of a batch verification, split by generator range over 2 ranks"""
    import random
    from bulletproofs_gadgets_amd import r1cs as g
    rnd = random.Random(6)
    pts = [(sonny.random_point(rnd), sonny.random_point(rnd)) for _ in range(3)]
    gbe = GpuBackend(gpu_ctx)
    proofs = [rc.addition_roundtrip(gbe, a, b, a + b, rng=R.SeededRng(b"sb%d" % i), cap=64)[1] for i, (a, b) in enumerate(pts)]

    def fn(be):
        bp = be.bp_gens(64)

        def verifiers():
            vs = []
            for a, b in pts:
                v = be.Verifier(b"Point Addition")
                g1, g2, g3 = (be.EdwardsGadget.from_point(p) for p in (a, b, a + b))
                g3.equal(g1.add(g2, v), v)
                vs.append(v)
            return vs
        ok = g.verify_batch(verifiers(), proofs, bp, bytes(range(64)) * 3)
        bad = list(proofs)
        bad[1] = proofs[0]
        try:
            g.verify_batch(verifiers(), bad, bp, bytes(range(64)) * 3)
            rejected = False
        except g.VerificationError:
            rejected = True
        return ok, rejected
    assert run_ranks(gpu_ctx, 2, fn) == [(True, True)] * 2


def test_sharded_prove_many(gpu_ctx):
    """bpg_prover_prove_many on sharded generators: the 3g- and 2g-target MSMs of a group of proofs are split by
    generator range and their partial sums exchanged in one all-gather per MSM; every rank gets the bytes of the
    unsharded prover's separate prove() calls"""
    import hashlib
    import random
    from bulletproofs_gadgets_amd import r1cs as g
    rnd = random.Random(9)
    wit = [(sonny.random_point(rnd), sonny.random_point(rnd)) for _ in range(3)]
    ext = b"".join(hashlib.sha256(b"spm%d" % j).digest() for j in range(3))

    def build(be, a, b):
        p = be.Prover(b"Point Addition")
        g1, g2, g3 = (be.EdwardsGadget.from_point(x) for x in (a, b, a + b))
        g3.equal(g1.add(g2, p), p)
        return p
    gbe = GpuBackend(gpu_ctx)
    bp = gbe.bp_gens(128)
    want = [build(gbe, *wit[j]).prove(bp, ext[32 * j:32 * j + 32]).to_bytes() for j in range(3)]

    def fn(be):
        be.bp_gens(128)
        return [p.to_bytes() for p in g.Prover.prove_many([build(be, *w) for w in wit], ext)]
    for got in run_ranks(gpu_ctx, 2, fn):
        assert got == want


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_two_phase_proof(gpu_ctx, oracle, world):
    """randomized (two-phase) constraints over sharded generators: the first- and second-phase commitments are MSMs
    over sub-ranges [0, n1) and [n1, n) of G / H that cut across the ranks' ranges; bytes = the oracle's"""
    import random
    rnd = random.Random(99)
    L = sonny.L
    xs = [rnd.randrange(L) for _ in range(9)]
    perm = xs[3:] + xs[:3]
    want_ok, want = rc.shuffle_roundtrip(OracleBackend(oracle), xs, perm, first_phase_squares=5, rng=R.SeededRng(b"sh2p"), cap=64)
    assert want_ok

    def fn(be):
        ok, proof = rc.shuffle_roundtrip(be, xs, perm, first_phase_squares=5, rng=R.SeededRng(b"sh2p"), cap=64)
        return ok, proof.to_bytes()
    for ok, raw in run_ranks(gpu_ctx, world, fn):
        assert ok and raw == want.to_bytes()
