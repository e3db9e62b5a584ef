"""Host time per MSM: submit (launches), wait on a finished job (host Horner over the window sums), combine."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bulletproofs_gadgets_amd as bpg  # noqa: E402
from bulletproofs_gadgets_amd import synth, _native  # noqa: E402


def main():
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n = 1 << logn
    ctx = bpg.Context(0)
    pts = ctx.points_from_uniform(synth.uniform_bytes(n))
    sc = synth.scalars(n)
    buf = ctx.alloc(len(sc)).upload(sc)
    ctx.msm_submit(pts, buf.ptr, n).wait()
    sub, wait, comb = [], [], []
    for _ in range(20):
        t0 = time.perf_counter()
        j = ctx.msm_submit(pts, buf.ptr, n)
        t1 = time.perf_counter()
        time.sleep(0.01)
        t2 = time.perf_counter()
        part = j.wait_partial()
        t3 = time.perf_counter()
        _native.combine_partials(part)
        t4 = time.perf_counter()
        sub.append(t1 - t0)
        wait.append(t3 - t2)
        comb.append(t4 - t3)
    med = lambda v: round(sorted(v)[len(v) // 2] * 1e6, 1)
    print(json.dumps({"log2n": logn, "submit_us": med(sub), "wait_finished_us": med(wait), "combine_us": med(comb)}))


if __name__ == "__main__":
    main()
