"""MSM timing on skewed scalar distributions (gadget witnesses are mostly bits and small values)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bulletproofs_gadgets_amd as bpg  # noqa: E402
from bulletproofs_gadgets_amd import synth  # noqa: E402


def main():
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n = 1 << logn
    ctx = bpg.Context(0)
    pts = ctx.points_from_uniform(synth.uniform_bytes(n))
    rng = np.random.default_rng(1)
    cases = {}
    u = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    u[:, 31] &= 0x0f
    cases["uniform"] = u
    b = np.zeros((n, 32), dtype=np.uint8)
    b[:, 0] = rng.integers(0, 2, size=n, dtype=np.uint8)
    cases["bits"] = b
    s = np.zeros((n, 32), dtype=np.uint8)
    s[:, :3] = rng.integers(0, 256, size=(n, 3), dtype=np.uint8)
    s[:, 2] &= 0x0f
    cases["below_2^20"] = s
    e = np.zeros((n, 32), dtype=np.uint8)
    e[:] = u[0]
    cases["all_equal"] = e
    m = u.copy()
    m[::2] = b[::2]
    cases["half_bits_half_uniform"] = m
    for name, arr in cases.items():
        sc = arr.tobytes()
        buf = ctx.alloc(len(sc)).upload(sc)
        ctx.msm_submit(pts, buf.ptr, n).wait()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            r = ctx.msm_submit(pts, buf.ptr, n).wait()
            ts.append(time.perf_counter() - t0)
        ka, kt = ctx.msm_last_kernel_us()
        print(json.dumps({"case": name, "log2n": logn, "wall_ms": round(min(ts) * 1e3, 3), "device_total_us": round(kt, 1),
                          "accumulate_us": round(ka, 1), "out": r.hex()[:16]}), flush=True)
        buf.free()


if __name__ == "__main__":
    main()
