"""Randomised parity soak on the GPU (not a test: run by hand, `python3 tools/soak.py [seconds]`).
MSMs of random sizes over random and adversarial scalars against the oracle's bytes, and small two-phase / one-phase
proofs against the oracle's proof bytes, until the time is up.  Prints one JSON line."""
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bulletproofs_gadgets_amd as bpg  # noqa: E402
from bulletproofs_gadgets_amd import synth  # noqa: E402
from oracle import binding, r1cs as R, sonny  # noqa: E402  (checker)
from tests import ref_cases as rc  # noqa: E402
from tests.backends import GpuBackend, OracleBackend  # noqa: E402

L = synth.L


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rnd = random.Random(seed)
    orc = binding.load()
    ctx = bpg.Context(0)
    NMAX = 24000
    u = synth.uniform_bytes(NMAX, tag=b"soak%d" % seed)
    pts = ctx.points_from_uniform(u)
    arr, _ = orc.from_uniform(u)
    edge = [0, 1, 2, L - 1, L - 2, (L - 1) // 2, (L + 1) // 2, 1 << 252, (1 << 252) - 1, 1 << 15, 1 << 13, (1 << 14) - 1,
            0x8000800080008000800080008000, sum(1 << (14 * k + 13) for k in range(18)) % L, sum(1 << (16 * k + 15) for k in range(15))]
    gbe, obe = GpuBackend(ctx), OracleBackend(orc)
    t0 = time.time()
    n_msm = n_proof = pts_total = 0
    last = t0
    while time.time() - t0 < budget:
        if time.time() - last > 20:                         # a progress line now and then (long silent runs look hung)
            last = time.time()
            print("soak: %d MSMs, %d proofs, %.0f s" % (n_msm, n_proof, last - t0), flush=True)
        kind = rnd.random()
        n = rnd.choice([rnd.randrange(1, 64), rnd.randrange(64, 2000), rnd.randrange(2000, NMAX)])
        if kind < 0.5:
            sc = b"".join(rnd.randrange(L).to_bytes(32, "little") for _ in range(n))
        elif kind < 0.75:
            sc = b"".join(rnd.choice(edge).to_bytes(32, "little") for _ in range(n))
        elif kind < 0.9:
            sc = b"".join(rnd.randrange(1 << rnd.choice([1, 8, 20, 64])).to_bytes(32, "little") for _ in range(n))
        else:                                               # unreduced 256-bit inputs are taken mod l by both sides
            sc = b"".join(((rnd.getrandbits(256)) % L).to_bytes(32, "little") for _ in range(n))
        got = ctx.msm(pts, sc)
        want = orc.msm(sc, arr, n)
        assert got == want, ("msm", seed, n, kind)
        n_msm += 1
        pts_total += n
        if n_msm % 8 == 0:                                  # a proof now and then: shuffle (two-phase) or scalar_mul
            tag = b"soak-%d-%d" % (seed, n_proof)
            if rnd.random() < 0.5:
                k = rnd.randrange(2, 24)
                xs = [rnd.randrange(L) for _ in range(k)]
                ys = xs[:]
                rnd.shuffle(ys)
                sq = rnd.randrange(0, 4)
                a = rc.shuffle_roundtrip(gbe, xs, ys, first_phase_squares=sq, rng=R.SeededRng(tag), cap=128)
                b = rc.shuffle_roundtrip(obe, xs, ys, first_phase_squares=sq, rng=R.SeededRng(tag), cap=128)
            else:
                bits = [rnd.randrange(2) for _ in range(rnd.randrange(1, 14))]
                base = sonny.random_point(rnd)
                exp = base * sum(b_ << i for i, b_ in enumerate(bits))
                a = rc.scalar_mul_roundtrip(gbe, base, bits, exp, rng=R.SeededRng(tag))
                b = rc.scalar_mul_roundtrip(obe, base, bits, exp, rng=R.SeededRng(tag))
            assert a[0] and b[0] and a[1].to_bytes() == b[1].to_bytes(), ("proof", seed, n_proof)
            n_proof += 1
    print(json.dumps({"seed": seed, "seconds": round(time.time() - t0, 1), "msms": n_msm, "points": pts_total, "proofs": n_proof,
                      "mismatches": 0}))
    ctx.close()


if __name__ == "__main__":
    main()
