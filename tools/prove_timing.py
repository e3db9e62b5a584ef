"""R1CS proof timing on one GPU (BASELINE config 3: Edwards scalar_mul gadget with
2427 committed bits = 65 529 multipliers -> n+ = 2^16; prove + verify)."""
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bulletproofs_gadgets_amd as bpg  # noqa: E402
from bulletproofs_gadgets_amd import r1cs as g  # noqa: E402
from oracle import sonny  # noqa: E402  (witness values only: Sonny curve arithmetic in Python ints)
from oracle.r1cs import SeededRng  # noqa: E402


def main():
    nbits = int(sys.argv[1]) if len(sys.argv) > 1 else 2427
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rnd = random.Random(1)
    base = sonny.random_point(rnd)
    bits = [rnd.randrange(2) for _ in range(nbits)]
    k = sum(b << i for i, b in enumerate(bits))
    expected = base * k
    ctx = bpg.Context(0)
    n = 27 * nbits + 4
    cap = 1 << (n - 1).bit_length()
    t0 = time.perf_counter()
    gens = g.BulletproofGens(ctx, cap, 1)
    ctx.synchronize()
    print(json.dumps({"gens_capacity": cap, "gens_create_s": round(time.perf_counter() - t0, 3)}), flush=True)
    for r in range(reps):
        rng = SeededRng(b"timing%d" % r)
        bl = [rng.scalar() for _ in range(nbits)]
        t0 = time.perf_counter()
        p = g.Prover(gens, b"ScalarMul")
        comms, vars_ = p.commit_batch(bits, bl)
        t1 = time.perf_counter()
        q = g.SonnyEdwardsPointGadget.scalar_mul(g.SonnyEdwardsPointGadget.from_point(base), vars_, p)
        q.equal(g.SonnyEdwardsPointGadget.from_point(expected), p)
        t2 = time.perf_counter()
        proof = p.prove(gens, rng.fill_bytes(32))
        t3 = time.perf_counter()
        v = g.Verifier(b"ScalarMul")
        vv = [v.commit(c) for c in comms]
        q2 = g.SonnyEdwardsPointGadget.scalar_mul(g.SonnyEdwardsPointGadget.from_point(base), vv, v)
        q2.equal(g.SonnyEdwardsPointGadget.from_point(expected), v)
        t4 = time.perf_counter()
        ok = v.verify(proof, None, gens, rng.fill_bytes(32))
        t5 = time.perf_counter()
        print(json.dumps({"multipliers": p.stats()[0], "constraints": p.stats()[1], "proof_bytes": len(proof.to_bytes()),
                          "commit_ms": round((t1 - t0) * 1e3, 2), "synthesis_ms": round((t2 - t1) * 1e3, 2),
                          "prove_ms": round((t3 - t2) * 1e3, 2), "prove_breakdown": {k_: round(v_, 2) for k_, v_ in p.timing().items()},
                          "verifier_synthesis_ms": round((t4 - t3) * 1e3, 2), "verify_ms": round((t5 - t4) * 1e3, 2), "ok": ok}),
              flush=True)


if __name__ == "__main__":
    main()
