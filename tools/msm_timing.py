"""Quick MSM timing sweep on one GPU (development aid; bench.py is the contract)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bulletproofs_gadgets_amd as bpg  # noqa: E402
from bulletproofs_gadgets_amd import synth  # noqa: E402


def main():
    sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [16, 18, 20]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    ctx = bpg.Context(0)
    nmax = 1 << max(sizes)
    t0 = time.time()
    u = synth.uniform_bytes(nmax)
    sc = synth.scalars(nmax)
    t1 = time.time()
    pts = ctx.points_from_uniform(u)
    ctx.synchronize()
    t2 = time.time()
    print(json.dumps({"gen_inputs_s": round(t1 - t0, 2), "table_build_s": round(t2 - t1, 3), "n": nmax}), flush=True)
    buf = ctx.alloc(len(sc)).upload(sc)
    for lg in sizes:
        n = 1 << lg
        plan = ctx.msm_plan(n)
        ctx.msm_submit(pts, buf.ptr, n).wait()  # warm (allocations)
        ts, acc, tot = [], [], []
        for _ in range(reps):
            a = time.time()
            r = ctx.msm_submit(pts, buf.ptr, n).wait()
            ts.append(time.time() - a)
            ka, kt = ctx.msm_last_kernel_us()
            acc.append(ka)
            tot.append(kt)
        # pipelined: 2 in flight
        a = time.time()
        jobs = []
        K = 2 * reps
        for i in range(K):
            jobs.append(ctx.msm_submit(pts, buf.ptr, n))
            if len(jobs) == 2:
                jobs.pop(0).wait()
        for j in jobs:
            j.wait()
        pipe = (time.time() - a) / K
        print(json.dumps({"log2n": lg, "plan": plan, "wall_ms_min": round(min(ts) * 1e3, 3),
                          "device_total_us_min": round(min(tot), 1), "accumulate_us_min": round(min(acc), 1),
                          "pipelined_ms_per_msm": round(pipe * 1e3, 3),
                          "Mpts_per_s_pipelined": round(n / pipe / 1e6, 1), "out": r.hex()[:16]}), flush=True)


if __name__ == "__main__":
    main()
