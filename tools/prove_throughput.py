"""R1CS proofs/s on one GPU with T host threads (BASELINE config 3 circuit: Edwards scalar_mul
gadget with `nbits` committed bits = 27*nbits+4 multipliers).  Each thread drives its own
proofs on its own stream against one shared generator object (one proof = one thread, as in
the reference); `mode` = prove | prove+verify."""
import json
import os
import random
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bulletproofs_gadgets_amd as bpg  # noqa: E402
from bulletproofs_gadgets_amd import r1cs as g  # noqa: E402
from oracle import sonny  # noqa: E402  (witness values only: Sonny curve arithmetic in Python ints)
from oracle.r1cs import SeededRng  # noqa: E402


def run(ctx, gens, nbits, threads, per_thread, verify):
    rnd = random.Random(1)
    base = sonny.random_point(rnd)
    bits = [rnd.randrange(2) for _ in range(nbits)]
    expected = base * sum(b << i for i, b in enumerate(bits))
    E = g.SonnyEdwardsPointGadget
    oks, errs = [], []

    import hashlib
    vals = b"".join(int(b).to_bytes(32, "little") for b in bits)

    def one(tid, r):
        seed = b"tp%d-%d" % (tid, r)
        # blinding factors: 31 uniform bytes each (< 2^248 < l), one SHAKE call per proof
        raw = hashlib.shake_256(seed).digest(31 * nbits + 64)
        bl = b"".join(raw[31 * i:31 * i + 31] + b"\0" for i in range(nbits))
        p = g.Prover(gens, b"ScalarMul")
        comms, vars_ = p.commit_batch_raw(vals, bl)
        E.scalar_mul(E.from_point(base), vars_, p).equal(E.from_point(expected), p)
        proof = p.prove(gens, raw[-32:])
        if verify:
            v = g.Verifier(b"ScalarMul")
            vv = v.commit_batch_raw(comms.raw)
            E.scalar_mul(E.from_point(base), vv, v).equal(E.from_point(expected), v)
            oks.append(v.verify(proof, None, gens, raw[-64:-32]))

    def worker(tid):
        try:
            for r in range(per_thread):
                one(tid, r)
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))
    one(99, 0)                         # warm-up: workspace allocation on the main thread's lane
    th = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    assert not errs, errs
    assert all(oks)
    return threads * per_thread / dt, dt


def main():
    nbits = int(sys.argv[1]) if len(sys.argv) > 1 else 2427
    per_thread = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    tlist = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,2,4,8,16").split(",")]
    ctx = bpg.Context(0)
    n = 27 * nbits + 4
    cap = 1 << (n - 1).bit_length()
    gens = g.BulletproofGens(ctx, cap, 1)
    for verify in (False, True):
        for T in tlist:
            rate, dt = run(ctx, gens, nbits, T, per_thread, verify)
            print(json.dumps({"multipliers": n, "threads": T, "mode": "prove+verify" if verify else "prove",
                              "proofs": T * per_thread, "seconds": round(dt, 3), "proofs_per_s": round(rate, 2)}), flush=True)


if __name__ == "__main__":
    main()
