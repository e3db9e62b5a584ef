"""One oracle proof of bench.py's secondary-metric statement on this core (CPU only; bench.py starts one of these
per host core for the all-cores CPU baseline).  Prints one JSON line: wall-clock start / end of the proving work
(time.time, comparable across workers), the share of it spent inside oracle/*.c, and the proof's SHA-256."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class _TimedLib:
    """ctypes library proxy that adds up the time spent inside its functions"""

    def __init__(self, lib):
        self._lib, self.seconds, self._cache = lib, 0.0, {}

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not callable(fn):
            return fn
        w = self._cache.get(name)
        if w is None:
            def w(*a, _fn=fn):
                t0 = time.perf_counter()
                try:
                    return _fn(*a)
                finally:
                    self.seconds += time.perf_counter() - t0
            self._cache[name] = w
        return w


def main():
    nbits = int(sys.argv[1]) if len(sys.argv) > 1 else 2427
    from bench import proof_statement_bytes
    from oracle import binding, gadgets as OG, r1cs as OR
    orc = binding.load()
    base, bits, expected, vals, randomness = proof_statement_bytes(nbits)
    bl, ext_p, _ = randomness(b"parity")
    n = 27 * nbits + 4
    pc, bp = OR.PedersenGens(orc), OR.BulletproofGens(orc, 1 << (n - 1).bit_length(), 1)     # setup, untimed (as on the GPU side)
    timed = _TimedLib(orc.lib)
    orc.lib = timed
    start, t0 = time.time(), time.perf_counter()
    prover = OR.Prover(orc, pc, OR.Transcript(orc, b"ScalarMul"))
    vars_ = [prover.commit(b, int.from_bytes(bl[32 * i:32 * i + 32], "little"))[1] for i, b in enumerate(bits)]
    EG = OG.SonnyEdwardsPointGadget
    EG.scalar_mul(EG.from_point(base), vars_, prover).equal(EG.from_point(expected), prover)
    proof = prover.prove(bp, ext_p)
    dt = time.perf_counter() - t0
    print(json.dumps({"start": start, "end": start + dt, "seconds": dt, "c_frac": timed.seconds / dt,
                      "sha256": hashlib.sha256(proof.to_bytes()).hexdigest()}), flush=True)


if __name__ == "__main__":
    main()
