"""BASELINE config 5: a batch of K Ristretto-point-gadget proofs (P1 + P2 = P3, 134 multipliers ->
n+ = 256), proved by T host threads on one GPU, then verified (a) one by one and (b) by ONE
combined multiscalar multiplication (bpg_verifier_verify_batch).  Prints one JSON line."""
import hashlib
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bulletproofs_gadgets_amd as bpg  # noqa: E402
from bulletproofs_gadgets_amd import r1cs as g  # noqa: E402
from bulletproofs_gadgets_amd import synth  # noqa: E402


class _RP:
    """zerocaf RistrettoPoint(pub EdwardsPoint) as the binding expects it"""
    def __init__(self, p):
        self.p = p


def statements(k):
    """k statements (P1, P2, P1+P2) on the Sonny curve, derived from the bench seed"""
    base = synth.sonny_base()
    out, p = [], base
    for i in range(k):
        q = p + base          # walk the subgroup: cheap, deterministic, all distinct
        out.append((_RP(p), _RP(q), _RP(p + q)))
        p = q + p
    return out


def circuit(cs, st):
    R = g.SonnyRistrettoPointGadget
    g1, g2, g3 = (R.from_point(p, cs) for p in st)
    g3.equals(cs, g1.add(cs, g2))


def run(ctx, gens, k, threads, rank=0):
    sts = statements(k)
    proofs = [None] * k
    errs = []

    def prove_range(lo, hi):
        try:
            for i0 in range(lo, hi, 8):             # eight statements per bpg_prover_prove_many call
                idx = range(i0, min(hi, i0 + 8))
                ps = []
                for i in idx:
                    p = g.Prover(gens, b"RistrettoAdd")
                    circuit(p, sts[i])
                    ps.append(p)
                ext = b"".join(hashlib.sha256(b"bv%d/%d" % (rank, i)).digest() for i in idx)
                for i, proof in zip(idx, g.Prover.prove_many(ps, ext)):
                    proofs[i] = proof
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))
    prove_range(0, 1)
    per = (k + threads - 1) // threads
    th = [threading.Thread(target=prove_range, args=(t * per, min(k, (t + 1) * per))) for t in range(threads)]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    prove_s = time.perf_counter() - t0
    assert not errs, errs

    def verifiers(lo, hi):
        vs = []
        for i in range(lo, hi):
            v = g.Verifier(b"RistrettoAdd")
            circuit(v, sts[i])
            vs.append(v)
        return vs
    t0 = time.perf_counter()
    vs = verifiers(0, k)
    synth_s = time.perf_counter() - t0
    rnd = hashlib.shake_256(b"bv-rng").digest(64 * k)
    t0 = time.perf_counter()
    ok = g.verify_batch(vs, proofs, gens, rnd)
    batch_s = time.perf_counter() - t0
    # the same batch as `threads` sub-batches, one host thread and one combined MSM each
    vs = verifiers(0, k)
    oks = [False] * threads

    def verify_range(t, lo, hi):
        try:
            oks[t] = lo >= hi or g.verify_batch(vs[lo:hi], proofs[lo:hi], gens, rnd[64 * lo:64 * hi])
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))
    th = [threading.Thread(target=verify_range, args=(t, t * per, min(k, (t + 1) * per))) for t in range(threads)]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    batch_mt_s = time.perf_counter() - t0
    assert not errs, errs
    ok = bool(ok) and all(oks)
    ksingle = min(k, 256)
    vs = verifiers(0, ksingle)
    t0 = time.perf_counter()
    for i in range(ksingle):
        assert vs[i].verify(proofs[i], None, gens, bytes(32))
    single_s = (time.perf_counter() - t0) * k / ksingle
    # a corrupted proof anywhere must fail the batch
    bad = list(proofs)
    raw = bytearray(bad[k // 2].to_bytes())
    raw[-1] ^= 1
    bad[k // 2] = g.Proof(bytes(raw))
    try:
        g.verify_batch(verifiers(0, k), bad, gens, hashlib.shake_256(b"bv-rng2").digest(64 * k))
        rejected = False
    except (g.VerificationError, bpg.BpgError):
        rejected = True
    return {"proofs": k, "multipliers": 134, "prove_threads": threads, "prove_s": round(prove_s, 3),
            "proofs_per_s_prove": round(k / prove_s, 1), "verifier_synthesis_s": round(synth_s, 3),
            "verify_batch_s": round(batch_s, 3), "proofs_per_s_verify_batch": round(k / batch_s, 1),
            "verify_batch_threads_s": round(batch_mt_s, 3), "proofs_per_s_verify_batch_threads": round(k / batch_mt_s, 1),
            "verify_one_by_one_s": round(single_s, 3), "proofs_per_s_verify_single": round(k / single_s, 1),
            "batch_ok": bool(ok), "bad_batch_rejected": rejected}


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    ctx = bpg.Context(0)
    gens = g.BulletproofGens(ctx, 256, 1)
    print(json.dumps(run(ctx, gens, k, threads)), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
