"""ONE R1CS proof split by generator range over the ranks of a torch.distributed job
(BASELINE config 4: Edwards scalar_mul gadget, `nbits` committed bits = 27*nbits+4 multipliers;
38 836 bits -> 2^20 multipliers).  Every rank captures the same circuit and witness, holds its
range of G / H resident, runs each MSM of prove / verify over that range, and all-gathers the
128-byte partial sums (RCCL with --backend nccl, one GPU per rank; --backend gloo lets several
ranks share one GPU for rehearsal and tests).

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
      tools/sharded_proof.py --nbits 300 --backend gloo --check
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def statement(nbits):
    """statement, packed witness bytes and the randomness every rank must share"""
    from bulletproofs_gadgets_amd import synth
    base, bits, expected = synth.scalar_mul_statement(nbits)
    vals = b"".join(int(b).to_bytes(32, "little") for b in bits)
    raw = hashlib.shake_256(synth.SEED + b"rng/sharded").digest(31 * nbits + 64)
    bl = b"".join(raw[31 * i:31 * i + 31] + b"\0" for i in range(nbits))
    return base, expected, vals, bl, raw[-64:-32], raw[-32:]


def prove_and_verify(g, gens, nbits, stmt=None, verify=True, early=None):
    """returns (proof, seconds: capture+commit, prove, verify, prover timing dict).  early: hand the prover its external
    randomness right after the commitments, so that the blinding draws run beside the circuit capture
    (bpg_prover_begin_blindings; same proof bytes); default on, BPG_EARLY_BLINDINGS=0 turns it off."""
    base, expected, vals, bl, ext_p, ext_v = stmt or statement(nbits)
    if early is None:
        early = os.environ.get("BPG_EARLY_BLINDINGS", "1") != "0"
    E = g.SonnyEdwardsPointGadget
    t0 = time.perf_counter()
    p = g.Prover(gens, b"ScalarMul")
    comms, vars_ = p.commit_batch_raw(vals, bl)
    if early:
        p.begin_blindings(ext_p)
    E.scalar_mul(E.from_point(base), vars_, p).equal(E.from_point(expected), p)
    t1 = time.perf_counter()
    proof = p.prove(gens, ext_p)
    t2 = time.perf_counter()
    if verify:
        v = g.Verifier(b"ScalarMul")
        E.scalar_mul(E.from_point(base), v.commit_batch_raw(comms.raw), v).equal(E.from_point(expected), v)
        assert v.verify(proof, None, gens, ext_v)
    t3 = time.perf_counter()
    return proof, t1 - t0, t2 - t1, t3 - t2, p.timing()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nbits", type=int, default=38836)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--check", action="store_true", help="also prove unsharded on every rank and compare the bytes")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    ndev = torch.cuda.device_count()
    dev = local_rank % max(ndev, 1)
    if world > 1:
        if args.backend == "nccl":
            torch.cuda.set_device(dev)
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo")
    import bulletproofs_gadgets_amd as bpg
    from bulletproofs_gadgets_amd import r1cs as g
    ctx = bpg.Context(dev)
    n = 27 * args.nbits + 4
    cap = 1 << (n - 1).bit_length()
    shard = None
    if world > 1:
        shard = (rank, world, g.make_all_gather(dist, "cuda" if args.backend == "nccl" else None))
    t0 = time.perf_counter()
    gens = g.BulletproofGens(ctx, cap, 1, shard=shard)
    gens_s = time.perf_counter() - t0
    stmt = statement(args.nbits)
    best, runs = None, []
    for _ in range(args.repeat):
        if world > 1:
            dist.barrier()
        proof, cap_s, prove_s, verify_s, tm = prove_and_verify(g, gens, args.nbits, stmt)
        runs.append((round(cap_s, 4), round(prove_s, 4)))
        if best is None or cap_s + prove_s < best[0] + best[1]:
            best = (cap_s, prove_s, verify_s, tm)
    ok = True
    if args.check:
        full = g.BulletproofGens(ctx, cap, 1)
        ref, *_ = prove_and_verify(g, full, args.nbits, stmt)
        ok = ref.to_bytes() == proof.to_bytes()
    digest = hashlib.sha256(proof.to_bytes()).hexdigest()
    if world > 1:
        res = [None] * world
        dist.all_gather_object(res, (ok, digest))
        ok = all(o for o, _ in res) and len({d for _, d in res}) == 1
    if rank == 0:
        print(json.dumps({"multipliers": n, "gens_capacity": cap, "ranks": world, "backend": args.backend if world > 1 else None,
                          "gens_setup_s": round(gens_s, 3), "capture_commit_s": round(best[0], 4), "prove_s": round(best[1], 4),
                          "verify_s": round(best[2], 4), "prover_timing_ms": {k: round(v, 2) for k, v in best[3].items()},
                          "capture_and_prove_s_runs": runs, "proof_bytes": len(proof.to_bytes()), "proof_sha256": digest,
                          "all_ranks_agree" + ("_and_equal_unsharded" if args.check else ""): ok}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
