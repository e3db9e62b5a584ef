#!/usr/bin/env python3
"""bench.py -- headline benchmark: Ristretto MSM points/s at 2^20 (BASELINE.json config 2).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one 2^20-point multiscalar multiplication per GPU over a table and
scalars already resident in HBM.  With N GPUs the job is ONE MSM of N * 2^20
points split by point range (weak scaling): every rank runs Pippenger on its
range, the ranks all-gather their 128-byte partial sums over RCCL, and the
partials are added and compressed.  `value` = N * 2^20 * K / max-over-ranks time.

The JSON line also carries
  roofline     -- the bucket-accumulation kernel (k_accumulate) against the measured
                  v_mad_u64_u32 issue rate of the chip (integer VALU; no MFMA, not HBM)
  cpu_baseline -- the oracle's restatement of dalek's vartime Pippenger on host
                  cores, on a bounded sample (N = 1, rank 0 only); also used as a parity check.
"""
import argparse
import ctypes
import json
import os
import sys
import time

# More than two MSMs in flight need more hardware queues than HIP's default of 4: streams that share a queue
# run in submission order, and a stream parked on the accumulate gate then holds up the others behind it.
# 12: measured best for the MSM (0.97-0.99 ms against 1.00 at 8 and 1.01 at 16) and within 3 % of the best for proofs
# (4 / 8 / 12 / 16 / 24 queues: 498 / 540 / 553 / 570 / 436 proofs/s with 64 prover threads).
# (read by the HIP runtime when it starts, so set before anything loads it)
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("BPG_BENCH_HW_QUEUES", "12")
# a proof that is proved alone (the latency leg) times its scalar-vector kernels with HIP events: secondary.scalar_kernels
os.environ.setdefault("BPG_TIME_KERNELS", "1")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# v_mad_u64_u32 issues once per 4 cycles per SIMD (64 lanes): 256 CU x 4 SIMD x 64 lanes x 2.4 GHz / 4 = 39.3 T MAC/s,
# the same rate as the guide's fp64 FMA figure.  tools/ubench (profiles/r02_ubench.json, kernels of 7 ms) measures
# 37.1 T/s for it on MI355X, with its anchors at 73.0 of 78.6 T simple int32 ops/s and 36.6 of 39.3 T fp64 FMA/s;
# the nominal figure is the denominator (round 1 used a figure measured with 75-us kernels: 28.7 T/s, too low).
VMAD_PEAK_GMACS = 39321.6
VMAD_MEASURED_GMACS = 37062.1
# HBM-side bytes of one k_accumulate launch at 2^20 points, from rocprofv3 PMC passes (FETCH_SIZE doubled per the
# guide's gfx950 correction, WRITE_SIZE as read): profiles/r02_pmc_*.csv, re-measured in round 3 with the same result
# (profiles/r03_pmc_fetch_size.csv: 1.424 GB-units per launch; r03_pmc_write_size.csv: 0.156 GB).  bench.py cannot collect PMC
# itself (a counter run serialises the kernels): the figure is a constant here, refreshed by tools/refresh_profiles.sh.
ACC_TRAFFIC_BYTES_2_20 = 3.00e9          # 2 x 1.42 GB fetched + 0.16 GB written (gathered table lines: every point is read once per window)
# field mul = 72 MAC, mixed add = 7M = 504 MAC, canonical c = 16 -> 16 windows.
MAC_PER_POINT_ACCUMULATE = 16 * 504          # bucket accumulation only (the dominant kernel)
MAC_PER_POINT_MSM = 8640                     # W(2^20, 16), whole MSM
BYTES_PER_POINT = 128                        # 32 B scalar + 96 B affine-Niels point


def host_cores():
    """the cores this process may use (the GPU box gives one GPU's job 16 of the machine's cores)"""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, int(os.environ.get("BPG_BENCH_CPU_CORES", os.environ.get("OMP_NUM_THREADS", "16")))))


def cpu_baseline(ctx, pts, scalars, sample_n):
    """Oracle (CPU port of dalek's Pippenger) on the first sample_n points: one MSM on one core, then one MSM per
    core on every core at once (the C call releases the GIL).  Returns (points/s one core, ok, s, points/s all cores, cores, s)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import binding
    orc = binding.load()
    enc = pts.compress(0, sample_n)            # same points as the GPU table, as ristretto bytes
    arr = orc.ge_array(sample_n)
    workers = host_cores()
    chunk = (sample_n + workers - 1) // workers

    def dec(k):                                # setup (untimed): decode points for the CPU
        s, e = k * chunk, min(sample_n, (k + 1) * chunk)
        if s < e:
            rc = orc.lib.oracle_decode_many(ctypes.byref(arr, orc.GE * s), enc[32 * s:32 * e], ctypes.c_size_t(e - s))
            assert rc == 0
    with ThreadPoolExecutor(workers) as ex:
        list(ex.map(dec, range(workers)))
    sc = scalars[:32 * sample_n]
    t0 = time.perf_counter()
    want = orc.msm(sc, arr, sample_n, binding.SCHED_PIPPENGER)
    dt = time.perf_counter() - t0
    got = ctx.msm(pts, sc)                     # pts[0:sample_n] with the same scalars
    t0 = time.perf_counter()
    with ThreadPoolExecutor(workers) as ex:    # the same MSM once per core, all cores busy
        res = list(ex.map(lambda _: orc.msm(sc, arr, sample_n, binding.SCHED_PIPPENGER), range(workers)))
    dt_all = time.perf_counter() - t0
    return sample_n / dt, got == want and all(r == want for r in res), dt, workers * sample_n / dt_all, workers, dt_all


def proof_algorithmic_macs(lg):
    """Multiply-accumulates of ONE proof at 2^lg multipliers (padded), bucket accumulation only, for the schedule
    this library runs: every multiscalar multiplication priced at the canonical window for
    these sizes, c = 13 -> 20 windows x 504 MAC per point (W(2^16, 13) without the bucket reduction).
    Points: A_I, A_O, S = 5n + 3; T_1..T_6 = 10; inner-product rounds before the generators are folded
    (k0 = lg - 11 of them) read all 2n original generators each; the fold itself passes the 2n generators once;
    the remaining rounds read 2 n_k folded generators each.  The reference's schedule (dalek: Straus MSMs and a
    2-term multiplication per folded generator per round) costs about 2.9e10 MAC at lg = 16."""
    n = 1 << lg
    k0 = lg - 11 if lg >= 12 else 0
    pts = (5 * n + 3) + 10 + 2 * n * k0 + (2 * n if k0 else 0) + sum(2 * (n >> k) + 2 for k in range(k0, lg))
    return pts * 20 * 504, pts


def scalar_kernel_report(ktimes):
    """HBM GB/s of the scalar-vector kernels (l(x) / r(x) / t(x) algebra and the inner-product folding) of one
    2^16-multiplier proof proved alone, from HIP events on their dispatches (bpg_prover_kernel_timing; median of the
    single-proof legs), against both ceilings: bytes / 8 TB/s, and multiplications mod l (one Montgomery pass = 104
    v_mad_u64_u32) / the multiply-add peak.  `bound` names the larger of the two fractions."""
    if not ktimes:
        return None
    out = {}
    for name in ktimes[0]:
        us = sorted(k[name][0] for k in ktimes)[len(ktimes) // 2]
        _, nbytes, mm = ktimes[0][name]
        if us <= 0:
            continue
        gbps, gmacs = nbytes / us / 1e3, mm * 104 / us / 1e3
        out[name] = {"us": round(us, 1), "bytes": int(nbytes), "GBps": round(gbps, 1), "hbm_frac": round(gbps / 8000.0, 4),
                     "montmul": int(mm), "GMACps": round(gmacs, 1), "valu_mac_frac": round(gmacs / VMAD_PEAK_GMACS, 4),
                     "bound": "valu_int32 (mod-l multiplications)" if gmacs / VMAD_PEAK_GMACS > gbps / 8000.0 else "hbm"}
    out["note"] = ("one proof alone (the kernels of a group of eight are eight times as large); k_ipp_rows / k_ipp_fold: first inner-product "
                   "round; bytes = every input read once + every output written once")
    return out


PROOF_BITS = 2427        # 27 * 2427 + 4 = 65 533 multipliers -> n+ = 2^16


def proof_statement_bytes(nbits):
    """cfg3 statement and per-proof randomness as packed bytes (no per-proof Python loops)"""
    import hashlib
    from bulletproofs_gadgets_amd import synth
    base, bits, expected = synth.scalar_mul_statement(nbits)
    vals = b"".join(int(b).to_bytes(32, "little") for b in bits)

    def randomness(tag):
        raw = hashlib.shake_256(synth.SEED + b"rng/" + tag).digest(31 * nbits + 64)
        bl = b"".join(raw[31 * i:31 * i + 31] + b"\0" for i in range(nbits))    # blindings < 2^248 < l
        return bl, raw[-64:-32], raw[-32:]
    return base, bits, expected, vals, randomness


def proof_bench(ctx, rank, threads, per_thread, per_call, nbits=PROOF_BITS):
    """R1CS proofs at 2^16 multipliers on this rank's GPU: `threads` host threads, each on its own
    stream against one shared generator table, proving `per_call` statements per bpg_prover_prove_many
    call (their blinding RNGs advance in lock-step; the GPU part is one proof at a time per thread).
    Returns (proofs, seconds prove-only, seconds prove+verify, single-proof latencies in ms, a proof)."""
    import hashlib
    import threading
    from bulletproofs_gadgets_amd import r1cs as g
    base, bits, expected, vals, randomness = proof_statement_bytes(nbits)
    n = 27 * nbits + 4
    gens = g.BulletproofGens(ctx, 1 << (n - 1).bit_length(), 1)
    E = g.SonnyEdwardsPointGadget

    ktimes = []

    def one(tag, verify, early=False):
        bl, ext_p, ext_v = randomness(tag)
        p = g.Prover(gens, b"ScalarMul")
        comms, vars_ = p.commit_batch_raw(vals, bl)
        if early:                                      # the blinding RNG's chain runs beside the capture (bpg_prover_begin_blindings)
            p.begin_blindings(ext_p)
        E.scalar_mul(E.from_point(base), vars_, p).equal(E.from_point(expected), p)
        proof = p.prove(gens, ext_p)
        kt = p.kernel_timing()
        if kt:
            ktimes.append(kt)
        if verify:
            v = g.Verifier(b"ScalarMul")
            E.scalar_mul(E.from_point(base), v.commit_batch_raw(comms.raw), v).equal(E.from_point(expected), v)
            assert v.verify(proof, None, gens, ext_v)
        return proof

    def many(tags, verify, rnd=None):
        ps, meta = [], []
        for tag in tags:
            bl, ext_p, ext_v = rnd[tag] if rnd is not None else randomness(tag)
            p = g.Prover(gens, b"ScalarMul")
            comms, vars_ = p.commit_batch_raw(vals, bl)
            E.scalar_mul(E.from_point(base), vars_, p).equal(E.from_point(expected), p)
            ps.append(p)
            meta.append((comms, ext_p, ext_v))
        proofs = g.Prover.prove_many(ps, b"".join(m[1] for m in meta))
        del ps
        if verify:
            vs = []
            for (comms, _, ext_v), proof in zip(meta, proofs):
                v = g.Verifier(b"ScalarMul")
                E.scalar_mul(E.from_point(base), v.commit_batch_raw(comms.raw), v).equal(E.from_point(expected), v)
                if verify == "batch":
                    vs.append(v)
                else:
                    assert v.verify(proof, None, gens, ext_v)
            if vs:                                     # the call's proofs checked by one combined MSM, random weights
                assert g.verify_batch(vs, proofs, gens, b"".join(m[2] + hashlib.sha256(m[2]).digest() for m in meta))
        return proofs

    first = one(b"parity", True)                       # also the warm-up of the main thread's lane
    assert many([b"parity", b"parity2"], False)[0].to_bytes() == first.to_bytes()
    lat = []
    for verify in (False, True):
        t0 = time.perf_counter()
        one(b"lat%d" % verify, verify)
        lat.append((time.perf_counter() - t0) * 1e3)
    assert one(b"parity", False, early=True).to_bytes() == first.to_bytes()
    t0 = time.perf_counter()
    one(b"lat-early", False, early=True)
    lat.append((time.perf_counter() - t0) * 1e3)
    secs, cpu, cpu_sys = [], [], []
    for verify in (False, True, "batch"):
        errs = []

        # the statements' randomness (commitment blindings, external RNG bytes) is input data: derived before the clock
        # starts, as the points and scalars of the MSM are (the derivation is a Python loop that holds the GIL)
        rnd_of = {}

        def worker(tid, warm):
            try:
                if warm:                               # same call shape as the timed pass: the lane's buffers reach their final size
                    many([b"%d/%d/w%d" % (rank, tid, r) for r in range(per_call)], verify)
                    for r in range(per_thread):
                        tag = b"%d/%d/%d" % (rank, tid, r)
                        rnd_of[tag] = randomness(tag)
                for r0 in range(0, 0 if warm else per_thread, per_call):
                    many([b"%d/%d/%d" % (rank, tid, r) for r in range(r0, min(per_thread, r0 + per_call))], verify, rnd_of)
            except Exception as e:  # noqa: BLE001
                errs.append(repr(e))
        for warm in (True, False):                     # first pass: every thread's lane allocates its workspace
            th = [threading.Thread(target=worker, args=(t, warm)) for t in range(threads)]
            t0, c0, o0 = time.perf_counter(), time.process_time(), os.times()
            [t.start() for t in th]
            [t.join() for t in th]
            dt, dc, o1 = time.perf_counter() - t0, time.process_time() - c0, os.times()
            assert not errs, errs
        secs.append(dt)
        cpu.append(dc)                                 # host CPU seconds of the whole process (all threads) in the timed pass
        cpu_sys.append(o1.system - o0.system)          # of which in the kernel (page faults, pinning, driver calls)
    return threads * per_thread, secs, lat, first, (cpu, cpu_sys, ktimes)


def proof_cpu_baseline(nbits=PROOF_BITS):
    """The oracle's prover (dalek's schedules: Straus-CT MSMs, per-element 2-term generator
    folds) on the same statement and randomness, one host core; returns (seconds, proof bytes)."""
    from oracle import binding, gadgets as OG, r1cs as OR
    from bulletproofs_gadgets_amd import synth
    orc = binding.load()
    base, bits, expected, vals, randomness = proof_statement_bytes(nbits)
    bl, ext_p, _ = randomness(b"parity")
    n = 27 * nbits + 4
    pc, bp = OR.PedersenGens(orc), OR.BulletproofGens(orc, 1 << (n - 1).bit_length(), 1)
    t0 = time.perf_counter()
    prover = OR.Prover(orc, pc, OR.Transcript(orc, b"ScalarMul"))
    vars_ = [prover.commit(b, int.from_bytes(bl[32 * i:32 * i + 32], "little"))[1] for i, b in enumerate(bits)]
    EG = OG.SonnyEdwardsPointGadget
    EG.scalar_mul(EG.from_point(base), vars_, prover).equal(EG.from_point(expected), prover)
    proof = prover.prove(bp, ext_p)
    return time.perf_counter() - t0, proof.to_bytes()


def proof_cpu_baseline_all_cores(cores, nbits=PROOF_BITS):
    """one oracle proof per core on `cores` cores at once: child processes that never touch the GPU
    (python -m tools.cpu_proof_worker).  Returns (proofs/s, wall seconds, fraction of a worker's time spent inside
    the oracle's C library -- the rest is Python orchestration)."""
    import subprocess
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, "-m", "tools.cpu_proof_worker", str(nbits)], cwd=ROOT, stdout=subprocess.PIPE,
                              env=dict(os.environ, HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS="1"))
             for _ in range(cores)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    wall = time.perf_counter() - t0
    recs = [json.loads(o.decode().strip().splitlines()[-1]) for o, p in zip(outs, procs) if p.returncode == 0 and o.strip()]
    if len(recs) != cores:
        raise RuntimeError("cpu proof workers failed: %d of %d" % (cores - len(recs), cores))
    span = max(r["end"] for r in recs) - min(r["start"] for r in recs)      # excludes interpreter start-up
    return cores / span, wall, sum(r["c_frac"] for r in recs) / cores, len({r["sha256"] for r in recs}) == 1


def main():
    if os.environ.get("BPG_BENCH_SWITCHINTERVAL"):      # experiment: interpreter lock hand-over between the prover threads
        sys.setswitchinterval(float(os.environ["BPG_BENCH_SWITCHINTERVAL"]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--logn", type=int, default=int(os.environ.get("BPG_BENCH_LOGN", "20")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-proofs", action="store_true", help="skip the secondary metric (R1CS proofs/s at 2^16 multipliers)")
    ap.add_argument("--proof-threads", type=int, default=int(os.environ.get("BPG_BENCH_PROOF_THREADS", "64")))
    ap.add_argument("--proofs-per-thread", type=int, default=24, help="per host thread, in calls of --proofs-per-call (long enough for the threads to fall out of step: host and GPU phases of different threads then overlap)")
    ap.add_argument("--proofs-per-call", type=int, default=8, help="statements per bpg_prover_prove_many call (1: plain prove calls)")
    ap.add_argument("--no-big-proof", action="store_true", help="skip the 2^20-multiplier proof (BASELINE configs[3])")
    ap.add_argument("--no-batch", action="store_true", help="skip the 4096-proof batch verification (BASELINE configs[4])")
    ap.add_argument("--no-extras", action="store_true", help="skip the latency, PCIe-inclusive and strong-scaling legs (profiling runs)")
    ap.add_argument("--in-flight", type=int, default=int(os.environ.get("BPG_BENCH_IN_FLIGHT", "4")),
                    help="MSM jobs submitted before the oldest is waited for (each on its own stream)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = torch = None
    if world > 1:
        import torch                           # torch first: its HIP runtime is the one both sides then share
        import torch.distributed as dist
        # BPG_BENCH_BACKEND=gloo: rehearsal of the N > 1 path with several ranks sharing one GPU (CPU tensors in the
        # collectives); the real run is one rank per GPU over RCCL ("nccl")
        backend = os.environ.get("BPG_BENCH_BACKEND", "nccl")
        local_rank %= max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        cdev = "cuda" if backend == "nccl" else "cpu"

    import bulletproofs_gadgets_amd as bpg
    from bulletproofs_gadgets_amd import _native, synth

    n = 1 << args.logn
    ctx = bpg.Context(local_rank)
    # this rank's point range of the global N*n-point MSM
    u = synth.uniform_bytes(n, tag=b"P/%d" % rank)
    sc = synth.scalars(n, tag=b"s/%d" % rank)
    t0 = time.perf_counter()
    pts = ctx.points_from_uniform(u)
    ctx.synchronize()
    table_s = time.perf_counter() - t0
    d_sc = ctx.alloc(len(sc)).upload(sc)

    if world > 1:
        gather_in = torch.empty(128, dtype=torch.uint8, device=cdev)
        gather_out = torch.empty(128 * world, dtype=torch.uint8, device=cdev)

    def finish(job):
        part = job.wait_partial()
        if world > 1:
            gather_in.copy_(torch.frombuffer(bytearray(part), dtype=torch.uint8))
            dist.all_gather_into_tensor(gather_out, gather_in)
            part = gather_out.cpu().numpy().tobytes()
        return _native.combine_partials(part)

    def barrier():
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        ctx.synchronize()

    def run(steps, acc_us=None):
        inflight, out = [], None
        for _ in range(steps):
            inflight.append(ctx.msm_submit(pts, d_sc.ptr, n))
            if len(inflight) == args.in_flight:
                out = finish(inflight.pop(0))
                if acc_us is not None:
                    acc_us.append(ctx.msm_last_kernel_us()[0])
        for j in inflight:
            out = finish(j)
            if acc_us is not None:
                acc_us.append(ctx.msm_last_kernel_us()[0])
        return out

    run(args.warmup)
    alone_us = []
    for _ in range(3):                         # the dominant kernel with nothing else on the GPU (not in the timed region)
        finish(ctx.msm_submit(pts, d_sc.ptr, n))
        alone_us.append(ctx.msm_last_kernel_us()[0])
    acc_us = []
    barrier()
    t0 = time.perf_counter()
    result = run(args.steps, acc_us)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- one MSM at a time: latency (submit -> result on the host), scalars resident / uploaded by the call
    def median(xs):
        xs = sorted(xs)
        return xs[len(xs) // 2]
    lat_res, lat_host = [], []
    for _ in range(0 if args.no_extras else 7):
        t0 = time.perf_counter()
        finish(ctx.msm_submit(pts, d_sc.ptr, n))
        lat_res.append((time.perf_counter() - t0) * 1e3)
    if world == 1 and not args.no_extras:
        for _ in range(5):
            t0 = time.perf_counter()
            ctx.msm(pts, sc)
            lat_host.append((time.perf_counter() - t0) * 1e3)
    # ---- PCIe-inclusive rate: every step uploads its 32 MB of scalars from host memory (bpg_msm, what a drop-in
    #      for multiscalar_mul(host scalars, resident points) does), `in_flight` host threads each on its own lane
    pcie = None
    if world == 1 and not args.no_extras:
        import threading
        per = max(2, args.steps // args.in_flight)

        def up_worker(k):
            for _ in range(k):
                ctx.msm(pts, sc)
        for k in (1, per):
            th = [threading.Thread(target=up_worker, args=(k,)) for _ in range(args.in_flight)]
            ctx.synchronize()
            t0 = time.perf_counter()
            [t.start() for t in th]
            [t.join() for t in th]
            dt_up = time.perf_counter() - t0
        pcie = {"value": args.in_flight * per * n / dt_up, "unit": "points/s", "host_threads": args.in_flight, "msms": args.in_flight * per,
                "note": "bpg_msm: scalars in pageable host memory, uploaded by every call (33.5 MB per MSM); never `value`"}

    # ---- strong scaling: ONE MSM of 2^20 (and of 2^22) points split N ways by point range; `value` above is the weak figure
    strong = {}
    for logt in (() if args.no_extras else (20, 22)):
        nl = (1 << logt) // world
        if world == 1 and logt == args.logn:
            strong["2^%d" % logt] = {"points_per_s": None, "note": "N = 1: the headline configuration itself"}
            continue
        try:
            pts_s = ctx.points_from_uniform(synth.uniform_bytes(nl, tag=b"S%d/%d/%d" % (logt, world, rank)))
            sc_s = synth.scalars(nl, tag=b"t%d/%d/%d" % (logt, world, rank))
            d_s = ctx.alloc(len(sc_s)).upload(sc_s)

            def run_s(steps):
                inflight, out = [], None
                for _ in range(steps):
                    inflight.append(ctx.msm_submit(pts_s, d_s.ptr, nl))
                    if len(inflight) == args.in_flight:
                        out = finish(inflight.pop(0))
                for j in inflight:
                    out = finish(j)
                return out
            run_s(max(2, args.warmup))
            barrier()
            t0 = time.perf_counter()
            run_s(args.steps)
            barrier()
            dts = time.perf_counter() - t0
            if world > 1:
                tm = torch.tensor([dts], dtype=torch.float64, device=cdev)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                dts = float(tm.item())
            strong["2^%d" % logt] = {"points_per_s": nl * world * args.steps / dts, "ms_per_msm": dts / args.steps * 1e3,
                                     "points_per_gpu": nl, "scaling": "strong"}
            d_s.free()
            pts_s.free()
        except Exception as e:  # noqa: BLE001
            strong["2^%d" % logt] = {"error": repr(e)}

    # ---- secondary metric: R1CS proofs/s at 2^16 multipliers (independent proofs; ranks share nothing)
    proofs = None
    if not args.no_proofs:
        barrier()
        cnt, secs, lat, first_proof, cpu_s3 = proof_bench(ctx, rank, args.proof_threads, args.proofs_per_thread, max(1, args.proofs_per_call))
        barrier()
        if world > 1:
            tt = torch.tensor(secs, dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            secs = [float(x) for x in tt.tolist()]
        s_prove, s_pv, s_pvb = secs
        proofs = {"metric": "R1CS proofs/s at 2^16 multipliers", "unit": "proofs/s",
                  "prove": world * cnt / s_prove, "prove_and_verify": world * cnt / s_pv,
                  "prove_and_verify_batched": world * cnt / s_pvb,      # each call's proofs verified by one combined MSM
                  "multipliers": 27 * PROOF_BITS + 4, "proof_bytes": len(first_proof.to_bytes()),
                  "host_threads_per_gpu": args.proof_threads, "proofs_per_gpu": cnt, "proofs_per_call": args.proofs_per_call,
                  "single_proof_latency_ms": {"prove": lat[0], "prove_and_verify": lat[1], "prove_with_early_blinding_draws": lat[2],
                                              "note": "capture + prove of one statement with the GPU to itself; the third figure hands the "
                                                      "external randomness over after the commitments (bpg_prover_begin_blindings): same bytes"},
                  "host_cpu_ms_per_proof": {"prove": 1e3 * cpu_s3[0][0] / cnt, "prove_and_verify": 1e3 * cpu_s3[0][1] / cnt,
                                            "prove_kernel_mode": 1e3 * cpu_s3[1][0] / cnt,
                                            "note": "process CPU time (all host threads; waits poll with 20-100 us naps unless BPG_SYNC=spin / block) "
                                                    "of this rank's timed pass / proofs; the box gives a rank 16 cores"},
                  "scalar_kernels": scalar_kernel_report(cpu_s3[2]),
                  "workload": "Edwards scalar_mul gadget, %d committed bits (BASELINE configs[2]); independent proofs, "
                              "one per host thread, sharded across GPUs with no collective" % PROOF_BITS}

    # ---- BASELINE configs[3]: ONE 2^20-multiplier proof, generator range sharded over the ranks, partial sums
    #      all-gathered over RCCL (N = 1: unsharded, the scaling baseline)
    big = None
    if not args.no_big_proof:
        try:
            from tools import sharded_proof as sp
            from bulletproofs_gadgets_amd import r1cs as g
            nbits = int(os.environ.get("BPG_BENCH_BIG_BITS", "38836"))
            cap = 1 << (27 * nbits + 3).bit_length()
            shard = (rank, world, g.make_all_gather(dist, cdev if cdev == "cuda" else None)) if world > 1 else None
            t0 = time.perf_counter()
            gens = g.BulletproofGens(ctx, cap, 1, shard=shard)
            gens_s = time.perf_counter() - t0
            stmt = sp.statement(nbits)
            best = None
            for _ in range(3):
                barrier()
                proof, cap_s, prove_s, verify_s, tm = sp.prove_and_verify(g, gens, nbits, stmt)
                if best is None or cap_s + prove_s < best[0] + best[1]:
                    best = (cap_s, prove_s, verify_s, tm)
            gens.free()
            import hashlib
            big = {"workload": "Edwards scalar_mul gadget, %d committed bits -> %d multipliers (BASELINE configs[3]); every MSM of "
                               "prove/verify split by generator range over the ranks, 128-byte partial sums all-gathered" % (nbits, 27 * nbits + 4),
                   "ranks": world, "gens_setup_s": round(gens_s, 3), "capture_commit_s": round(best[0], 3), "prove_s": round(best[1], 4),
                   "verify_s": round(best[2], 4), "prover_timing_ms": {k: round(v, 1) for k, v in best[3].items()},
                   "proof_sha256": hashlib.sha256(proof.to_bytes()).hexdigest()}
            # what every rank repeats whatever N is: circuit capture + commitments, and the blinding draw (one
            # sequential Keccak chain; bit-exactness fixes its order).  The MSMs are what the ranks split.
            # The draws start right after the commitments, on a helper thread beside the capture
            # (bpg_prover_begin_blindings), so the floor is the longer of the two, not their sum.
            early = os.environ.get("BPG_EARLY_BLINDINGS", "1") != "0"
            chain = best[3].get("rng_ms", 0.0) / 1e3
            serial = max(best[0], chain) if early else best[0] + chain
            total = best[0] + best[1]
            big["early_blindings"] = early
            big["blinding_chain_s"] = round(chain, 4)
            big["host_serial_s"] = round(serial, 4)
            big["amdahl_bound"] = {"total_s_at_this_n": round(total, 4),
                                   "speedup_limit_8_ranks": round(total / (serial + (total - serial) / 8), 2),
                                   "note": "capture + commit and the blinding RNG (one sequential Keccak chain, beside the capture on a helper "
                                           "thread) run on every rank; only the remainder (MSMs, scalar-vector kernels) can shrink with N"}
        except Exception as e:  # noqa: BLE001  (the headline line must survive a failure here)
            big = {"error": repr(e)}
    # ---- BASELINE configs[4]: 4096 Ristretto-gadget proofs per GPU, verified by one combined MSM (no collective)
    batch = None
    if not args.no_batch:
        try:
            from tools import batch_verify as bv
            from bulletproofs_gadgets_amd import r1cs as g
            gens = g.BulletproofGens(ctx, 256, 1)
            barrier()
            batch = bv.run(ctx, gens, int(os.environ.get("BPG_BENCH_BATCH", "4096")), args.proof_threads, rank)
            barrier()
            gens.free()
            if world > 1:
                tt = torch.tensor([batch["prove_s"], batch["verify_batch_s"], batch["verify_one_by_one_s"], batch["verify_batch_threads_s"]],
                                  dtype=torch.float64, device=cdev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                ps, vb, vs_, vt = (float(x) for x in tt.tolist())
                k = batch["proofs"]
                batch.update({"proofs": k * world, "prove_s": round(ps, 3), "verify_batch_s": round(vb, 3), "verify_one_by_one_s": round(vs_, 3),
                              "verify_batch_threads_s": round(vt, 3),
                              "proofs_per_s_prove": round(k * world / ps, 1), "proofs_per_s_verify_batch": round(k * world / vb, 1),
                              "proofs_per_s_verify_batch_threads": round(k * world / vt, 1),
                              "proofs_per_s_verify_single": round(k * world / vs_, 1)})
            batch["workload"] = "P1 + P2 = P3 with SonnyRistrettoPointGadget (134 multipliers); proofs sharded across GPUs, one combined MSM per GPU"
        except Exception as e:  # noqa: BLE001
            batch = {"error": repr(e)}

    if rank == 0:
        plan = ctx.msm_plan(n)
        acc = sum(acc_us) / len(acc_us) if acc_us else float("nan")
        achieved = MAC_PER_POINT_ACCUMULATE * n / (acc * 1e-6) / 1e9        # GMAC/s, algorithmic
        line = {
            "metric": "Ristretto MSM points/s at 2^20",
            "value": world * n * args.steps / dt,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 limbs (mod 2^255-19, mod l)",
            "data": "synthetic (SHAKE256 uniform points and scalars, seed bpg-bench-v1)",
            "config": {"workload": "raw Ristretto255 MSM, 2^%d points per GPU, point-range sharded (BASELINE configs[1])" % args.logn,
                       "points_per_gpu": n, "window_bits": plan["window_bits"], "windows": plan["windows"],
                       "in_flight": args.in_flight, "table_build_s": round(table_s, 4), "result": result.hex()},
            "roofline": {"bound": "valu_int32 (v_mad_u64_u32 issue)", "kernel": "k_accumulate",
                         "achieved": achieved, "peak": VMAD_PEAK_GMACS, "unit": "GMAC/s", "frac": achieved / VMAD_PEAK_GMACS,
                         "peak_measured": VMAD_MEASURED_GMACS, "frac_of_measured": achieved / VMAD_MEASURED_GMACS,
                         "peak_note": "256 CU x 4 SIMD x 64 lanes x 2.4 GHz / 4 cycles per v_mad_u64_u32; tools/ubench measures 37.1 T/s (profiles/r02_ubench.json)",
                         "kernel_avg_us": acc, "kernel_alone_us": min(alone_us), "frac_alone": MAC_PER_POINT_ACCUMULATE * n / (min(alone_us) * 1e-6) / 1e9 / VMAD_PEAK_GMACS,
                         "note": "kernel_avg_us: timed region, %d MSMs in flight (their accumulate passes run one after another, the other kernels overlap them); kernel_alone_us: the same kernel with the GPU to itself" % args.in_flight,
                         "algorithmic_mac_per_point": MAC_PER_POINT_ACCUMULATE,
                         "executed_mac_per_point": plan["windows"] * 504,
                         "hbm_algorithmic_GBps": BYTES_PER_POINT * n / (acc * 1e-6) / 1e9, "hbm_peak_GBps": 8000.0,
                         "traffic": ACC_TRAFFIC_BYTES_2_20 if args.logn == 20 else None},
        }
        line["latency_ms"] = {"msm_resident_scalars": median(lat_res) if lat_res else None, "msm_host_scalars": median(lat_host) if lat_host else None,
                              "note": "one 2^%d-point MSM with the GPU to itself, submit -> 32 result bytes on the host" % args.logn}
        if pcie is not None:
            line["pcie_inclusive"] = pcie
        line["strong_scaling"] = strong
        if world == 1 and not args.no_cpu_baseline:
            sample = n                      # the whole workload: ~3 s of one core at 2^20
            v, ok, cpu_s, v_all, cores, cpu_all_s = cpu_baseline(ctx, pts, sc, sample)
            line["cpu_baseline"] = {"value": v_all, "unit": "points/s", "cores": cores, "kind": "port",
                                    "sample": "oracle msm_pippenger_vt (dalek vartime Pippenger, w=8) on the 2^%d points of the step, one "
                                              "such MSM per core on %d cores at once, %.1f s" % (sample.bit_length() - 1, cores, cpu_all_s),
                                    "single_core": {"value": v, "cores": 1, "seconds": round(cpu_s, 2)},
                                    "parity_with_gpu": bool(ok)}
        if proofs is not None:
            lgp = (27 * PROOF_BITS + 3).bit_length()
            macs, ppts = proof_algorithmic_macs(lgp)
            ach = proofs["prove"] * macs / 1e9
            proofs["roofline"] = {"bound": "valu_int32 (v_mad_u64_u32 issue)", "achieved": ach, "peak": VMAD_PEAK_GMACS * world, "unit": "GMAC/s",
                                  "frac": ach / (VMAD_PEAK_GMACS * world), "algorithmic_mac_per_proof": macs, "msm_points_per_proof": ppts,
                                  "reference_schedule_mac_per_proof": 2.88e10,
                                  "note": "whole prove() calls per second (host work included) x the bucket-accumulation MACs of the "
                                          "schedule this library runs (fold-free first rounds, one fold), every point priced at SURVEY 8d's canonical "
                                          "window for these sizes (c = 13: 20 windows x 504 MAC; the library's own tables use 14-bit windows "
                                          "and a 7-bit fold, so it executes fewer); the reference's own schedule would need 2.9e10 MAC per proof"}
            if world == 1 and not args.no_cpu_baseline:
                cpu_s, cpu_proof = proof_cpu_baseline()
                cores = host_cores()
                proofs["cpu_baseline"] = {"value": 1.0 / cpu_s, "unit": "proofs/s", "cores": 1, "kind": "port",
                                          "sample": "oracle prover (Straus-CT MSMs, per-element generator folds), the same "
                                                    "statement, 1 proof, %.1f s" % cpu_s,
                                          "proof_bytes_equal_gpu": cpu_proof == first_proof.to_bytes()}
                try:
                    v_all, wall, c_frac, same = proof_cpu_baseline_all_cores(cores)
                    proofs["cpu_baseline"].update({
                        "value": v_all, "cores": cores,
                        "sample": "oracle prover (Straus-CT MSMs, per-element generator folds), the same statement, one proof per "
                                  "core on %d cores at once, %.1f s wall" % (cores, wall),
                        "single_core": {"value": 1.0 / cpu_s, "cores": 1, "seconds": round(cpu_s, 1)},
                        "time_inside_the_c_library": round(c_frac, 3),
                        "note": "orchestrated from Python (oracle/r1cs.py); time_inside_the_c_library is the share of a worker's "
                                "proving time spent in oracle/*.c (MSMs, folds, scalar vectors), the rest is interpreter overhead "
                                "a C or Rust prover would not pay",
                        "workers_agree": bool(same)})
                except Exception as e:  # noqa: BLE001
                    proofs["cpu_baseline"]["all_cores_error"] = repr(e)
            line["secondary"] = proofs
        if big is not None:
            line["proof_2_20"] = big
        if batch is not None:
            line["batch_4096"] = batch
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
