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

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Measured on MI355X by tools/ubench (profiles/r01_ubench.json): sustained v_mad_u64_u32, all CUs.
VMAD_PEAK_GMACS = 26450.0
# field mul = 72 MAC, mixed add = 7M = 504 MAC, canonical c = 16 -> 16 windows.
MAC_PER_POINT_ACCUMULATE = 16 * 504          # bucket accumulation only (the dominant kernel)
MAC_PER_POINT_MSM = 8640                     # W(2^20, 16), whole MSM
BYTES_PER_POINT = 128                        # 32 B scalar + 96 B affine-Niels point


def cpu_baseline(ctx, pts, scalars, sample_n):
    """Oracle (CPU port of dalek's Pippenger) on the first sample_n points; returns (points/s, ok)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import binding
    orc = binding.load()
    enc = pts.compress(0, sample_n)            # same points as the GPU table, as ristretto bytes
    arr = orc.ge_array(sample_n)
    workers = min(16, os.cpu_count() or 1)
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
    return sample_n / dt, got == want, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--logn", type=int, default=int(os.environ.get("BPG_BENCH_LOGN", "20")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = torch = None
    if world > 1:
        import torch                           # torch first: its HIP runtime is the one both sides then share
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

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
        gather_in = torch.empty(128, dtype=torch.uint8, device="cuda")
        gather_out = torch.empty(128 * world, dtype=torch.uint8, device="cuda")

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
            if len(inflight) == 2:
                out = finish(inflight.pop(0))
                if acc_us is not None:
                    acc_us.append(ctx.msm_last_kernel_us()[0])
        for j in inflight:
            out = finish(j)
            if acc_us is not None:
                acc_us.append(ctx.msm_last_kernel_us()[0])
        return out

    run(args.warmup)
    acc_us = []
    barrier()
    t0 = time.perf_counter()
    result = run(args.steps, acc_us)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

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
                       "in_flight": 2, "table_build_s": round(table_s, 4), "result": result.hex()},
            "roofline": {"bound": "valu_int32 (v_mad_u64_u32 issue)", "kernel": "k_accumulate",
                         "achieved": achieved, "peak": VMAD_PEAK_GMACS, "unit": "GMAC/s", "frac": achieved / VMAD_PEAK_GMACS,
                         "kernel_avg_us": acc, "algorithmic_mac_per_point": MAC_PER_POINT_ACCUMULATE,
                         "executed_mac_per_point": plan["windows"] * 504,
                         "hbm_algorithmic_GBps": BYTES_PER_POINT * n / (acc * 1e-6) / 1e9, "hbm_peak_GBps": 8000.0,
                         "traffic": None},
        }
        if world == 1 and not args.no_cpu_baseline:
            sample = min(n, 1 << 18)
            v, ok, cpu_s = cpu_baseline(ctx, pts, sc, sample)
            line["cpu_baseline"] = {"value": v, "unit": "points/s", "cores": 1, "kind": "port",
                                    "sample": "oracle msm_pippenger_vt (dalek vartime Pippenger, w=8) on the first 2^%d points, %.1f s"
                                              % (sample.bit_length() - 1, cpu_s),
                                    "parity_with_gpu": bool(ok)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
