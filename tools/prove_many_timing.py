"""Where a thread's time goes when it proves k statements per bpg_prover_prove_many call (2^16 multipliers)."""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import bulletproofs_gadgets_amd as bpg  # noqa: E402
from bulletproofs_gadgets_amd import r1cs as g  # noqa: E402


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ctx = bpg.Context(0)
    base, bits, expected, vals, randomness = bench.proof_statement_bytes(bench.PROOF_BITS)
    n = 27 * bench.PROOF_BITS + 4
    gens = g.BulletproofGens(ctx, 1 << (n - 1).bit_length(), 1)
    E = g.SonnyEdwardsPointGadget
    res = {}

    def work(tid):
        rows = []
        for rep in range(3):
            t0, c0 = time.perf_counter(), time.thread_time()
            ps, ext = [], b""
            for j in range(k):
                bl, ext_p, _ = randomness(b"%d-%d-%d" % (tid, rep, j))
                p = g.Prover(gens, b"ScalarMul")
                _, vars_ = p.commit_batch_raw(vals, bl)
                E.scalar_mul(E.from_point(base), vars_, p).equal(E.from_point(expected), p)
                ps.append(p)
                ext += ext_p
            t1, c1 = time.perf_counter(), time.thread_time()
            if k == 1:
                ps[0].prove(gens, ext)
            else:
                g.Prover.prove_many(ps, ext)
            t2, c2 = time.perf_counter(), time.thread_time()
            tm = [p.timing() for p in ps]
            rows.append({"capture_ms_per_proof": round((t1 - t0) * 1e3 / k, 2), "prove_ms_per_proof": round((t2 - t1) * 1e3 / k, 2),
                         "capture_cpu_ms_per_proof": round((c1 - c0) * 1e3 / k, 2), "prove_cpu_ms_per_proof": round((c2 - c1) * 1e3 / k, 2),
                         "rng_ms": round(sum(t["rng_ms"] for t in tm) / k, 2), "commit_ms": round(sum(t["commit_ms"] for t in tm) / k, 2),
                         "poly_ms": round(sum(t["poly_ms"] for t in tm) / k, 2), "ipp_ms": round(sum(t["ipp_ms"] for t in tm) / k, 2)})
        res[tid] = rows[-1]

    t0, c0, o0 = time.perf_counter(), time.process_time(), os.times()
    th = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    [t.start() for t in th]
    [t.join() for t in th]
    dt, dc, o1 = time.perf_counter() - t0, time.process_time() - c0, os.times()
    nproofs = 3 * k * threads
    print(json.dumps({"k": k, "threads": threads, "proofs_per_s_overall": round(nproofs / dt, 1),
                      "process_cpu_ms_per_proof": round(dc * 1e3 / nproofs, 2), "process_sys_ms_per_proof": round((o1.system - o0.system) * 1e3 / nproofs, 2),
                      "thread0": res[0]}))
    gens.free()
    ctx.close()          # BPG_TRACE_HOST=1: the library prints its host-phase table here


if __name__ == "__main__":
    main()
