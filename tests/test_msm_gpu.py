"""GPU parity tests of the MSM path (BASELINE config 2: raw Ristretto255 MSM,
bit-exact vs the dalek schedules restated in oracle/msm.c).  Through the C ABI.

Integer work: the bar is byte equality of the 32-byte compressed result.
"""
import hashlib

import pytest

from bulletproofs_gadgets_amd import synth
from bulletproofs_gadgets_amd._native import BpgError, EFORMAT

pytestmark = pytest.mark.gpu

L = synth.L


def _scalars_from_ints(vals):
    return b"".join(int(v % L).to_bytes(32, "little") for v in vals)


@pytest.fixture(scope="module")
def table(gpu_ctx, oracle):
    """4096 uniform points resident on the GPU + the same points in the oracle."""
    n = 4096
    u = synth.uniform_bytes(n, tag=b"P")
    pts = gpu_ctx.points_from_uniform(u)
    arr, _ = oracle.from_uniform(u)
    return pts, arr, n


def test_elligator_table_matches_oracle(gpu_ctx, oracle, table):
    pts, arr, n = table
    assert pts.compress(0, n) == oracle.encode(arr, n)


def test_decode_table_roundtrip(gpu_ctx, oracle, table):
    pts, arr, n = table
    enc = oracle.encode(arr, 512)
    t2 = gpu_ctx.points_from_compressed(enc)
    assert t2.compress() == enc


def test_decode_rejects_invalid(gpu_ctx):
    p = 2**255 - 19
    good = bytes(32)  # identity encoding is valid
    for bad in [p.to_bytes(32, "little"), (1).to_bytes(32, "little"), bytes([0xff] * 32), (2**255).to_bytes(32, "little")]:
        with pytest.raises(BpgError) as e:
            gpu_ctx.points_from_compressed(good + bad + good)
        assert e.value.code == EFORMAT


@pytest.mark.parametrize("n", [0, 1, 2, 3, 7, 64, 189, 190, 499, 500, 799, 800, 1024, 4096])
def test_msm_matches_oracle(gpu_ctx, oracle, table, n):
    pts, arr, _ = table
    sc = synth.scalars(n, tag=b"s%d" % n)
    got = gpu_ctx.msm(pts, sc)
    want = oracle.msm(sc, arr, n)
    assert got == want


def test_msm_edge_scalars(gpu_ctx, oracle, table):
    pts, arr, _ = table
    vals = [0, 1, L - 1, 2, L - 2, 2**252, 2**252 - 1, (L - 1) // 2, (L + 1) // 2, 2**13, 2**13 - 1, 2**14, 2**64 - 1,
            0, 0, 1, 1] + [2**k for k in range(0, 252, 9)] + [2**k - 1 for k in range(1, 252, 11)]
    sc = _scalars_from_ints(vals)
    n = len(vals)
    assert gpu_ctx.msm(pts, sc) == oracle.msm(sc, arr, n)
    # all-zero scalars -> identity
    assert gpu_ctx.msm(pts, bytes(32 * 100)) == bytes(32)


def test_msm_repeated_and_negated_points(gpu_ctx, oracle):
    # P, P, -P style cancellations: s*P + (l-s)*P = identity
    u = synth.uniform_bytes(1, tag=b"rep") * 8
    pts = gpu_ctx.points_from_uniform(u)
    s = int.from_bytes(hashlib.sha256(b"s").digest(), "little") % L
    sc = _scalars_from_ints([s, L - s, 5, L - 5, 0, 0, 7, L - 7])
    assert gpu_ctx.msm(pts, sc) == bytes(32)
    arr, _ = oracle.from_uniform(u)
    sc2 = _scalars_from_ints([s, s, 5, 5, 1, 2, 3, 4])
    assert gpu_ctx.msm(pts, sc2) == oracle.msm(sc2, arr, 8)


def test_msm_skewed_witness_scalars(gpu_ctx, oracle, table):
    """a_L / a_R in the gadget circuits are mostly bits and small values."""
    pts, arr, n = table
    sc = synth.small_scalars(n, 1, tag=b"bits")
    assert gpu_ctx.msm(pts, sc) == oracle.msm(sc, arr, n)
    sc = synth.small_scalars(n, 20, tag=b"small")
    assert gpu_ctx.msm(pts, sc) == oracle.msm(sc, arr, n)


def test_msm_multi_target_segments(gpu_ctx, oracle, table):
    pts, arr, n = table
    # 3 targets over segments [100,100+300) ++ [2000,2000+212) ++ [4095,4096)
    segs = [(100, 300), (2000, 212), (4095, 1)]
    m = sum(l for _, l in segs)
    sc = synth.scalars(3 * m, tag=b"mt")
    buf = gpu_ctx.alloc(len(sc)).upload(sc)
    got = gpu_ctx.msm_submit(pts, buf.ptr, m, 3, segments=segs).wait()
    sub, cnt = oracle.concat(*[(oracle.slice(arr, s, l), l) for s, l in segs])
    for t in range(3):
        want = oracle.msm(sc[32 * m * t:32 * m * (t + 1)], sub, cnt)
        assert got[32 * t:32 * t + 32] == want


def test_msm_two_jobs_in_flight(gpu_ctx, oracle, table):
    pts, arr, n = table
    s1, s2 = synth.scalars(n, tag=b"j1"), synth.scalars(n, tag=b"j2")
    b1, b2 = gpu_ctx.alloc(len(s1)).upload(s1), gpu_ctx.alloc(len(s2)).upload(s2)
    j1 = gpu_ctx.msm_submit(pts, b1.ptr, n)
    j2 = gpu_ctx.msm_submit(pts, b2.ptr, n)
    assert j2.wait() == oracle.msm(s2, arr, n)
    assert j1.wait() == oracle.msm(s1, arr, n)


def test_msm_linearity_at_size(gpu_ctx):
    """Size-independent property at 2^16: MSM(a) + MSM(b) == MSM(a+b) via a 2-point MSM of the results."""
    n = 1 << 16
    pts = gpu_ctx.points_from_uniform(synth.uniform_bytes(n, tag=b"big"))
    a, b = synth.scalars(n, tag=b"la"), synth.scalars(n, tag=b"lb")
    ab = b"".join(((int.from_bytes(a[32 * i:32 * i + 32], "little") + int.from_bytes(b[32 * i:32 * i + 32], "little")) % L)
                  .to_bytes(32, "little") for i in range(n))
    ra, rb, rab = gpu_ctx.msm(pts, a), gpu_ctx.msm(pts, b), gpu_ctx.msm(pts, ab)
    two = gpu_ctx.points_from_compressed(ra + rb)
    one = (1).to_bytes(32, "little")
    assert gpu_ctx.msm(two, one + one) == rab


@pytest.mark.parametrize("logn", [20, 22])
def test_msm_full_size_properties(gpu_ctx, logn):
    """BASELINE config 2 at its full sizes (2^20, 2^22), where the oracle would take minutes:
    size-independent properties instead.  (a) additivity over a point-range split, which is also the
    multi-GPU decomposition: MSM(all) == combine(partial(first 3/8), partial(rest)); (b) MSM(s) + MSM(l - s)
    == identity; (c) a sparse scalar vector (one non-zero entry per 2^14) equals the small MSM over exactly
    those points, which the oracle-checked sizes cover."""
    import numpy as np
    from bulletproofs_gadgets_amd import _native
    n = 1 << logn
    pts = gpu_ctx.points_from_uniform(synth.uniform_bytes(n, tag=b"full%d" % logn))
    rng = np.random.default_rng(logn)
    raw = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    raw[:, 31] &= 0x0f                                   # < 2^252 < l: canonical
    sc = raw.tobytes()
    buf = gpu_ctx.alloc(len(sc)).upload(sc)
    full = gpu_ctx.msm_submit(pts, buf.ptr, n).wait()
    assert full != bytes(32)
    # (a) range split 3/8 : 5/8 through the segment API and the partial-sum combine
    cut = 3 * n // 8
    p1 = gpu_ctx.msm_submit(pts, buf.ptr, cut, 1, segments=[(0, cut)]).wait_partial()
    p2 = gpu_ctx.msm_submit(pts, buf.ptr + 32 * cut, n - cut, 1, segments=[(cut, n - cut)]).wait_partial()
    assert _native.combine_partials(p1 + p2) == full
    # (b) negated scalars cancel: l - s, computed with numpy on 4 x u64 limbs
    limbs = raw.view("<u8").reshape(n, 4)
    neg = np.zeros((n, 4), dtype=np.uint64)
    borrow = np.zeros(n, dtype=bool)
    for k in range(4):
        lk = np.uint64((L >> (64 * k)) & (2**64 - 1))
        with np.errstate(over="ignore"):
            neg[:, k] = lk - limbs[:, k] - borrow.astype(np.uint64)
        borrow = (limbs[:, k] > lk) | ((limbs[:, k] == lk) & borrow)
    assert not borrow.any()
    negb = neg.astype("<u8").tobytes()
    nbuf = gpu_ctx.alloc(len(negb)).upload(negb)
    pn = gpu_ctx.msm_submit(pts, nbuf.ptr, n).wait_partial()
    pf = gpu_ctx.msm_submit(pts, buf.ptr, n).wait_partial()
    assert _native.combine_partials(pf + pn) == bytes(32)
    # (c) sparse vector == small dense MSM over the selected points
    step = 1 << 14
    idx = list(range(5, n, step))
    sparse = bytearray(32 * n)
    small = b""
    for j, i in enumerate(idx):
        s = synth.scalars(1, tag=b"sp%d" % j)
        sparse[32 * i:32 * i + 32] = s
        small += s
    sbuf = gpu_ctx.alloc(len(sparse)).upload(bytes(sparse))
    got = gpu_ctx.msm_submit(pts, sbuf.ptr, n).wait()
    sel = gpu_ctx.points_from_compressed(b"".join(pts.compress(i, 1) for i in idx))
    assert got == gpu_ctx.msm(sel, small)


def test_msm_full_size_skewed(gpu_ctx):
    """2^20 points with the digit distributions that fill single bins and buckets (the level-2 sort then runs
    many chunks per bin, the merge passes run deep): all scalars equal, all ones, random bits.
    MSM(s, s, ..., s) == s * MSM(1, ..., 1), the right side through a one-point MSM (an oracle-checked size);
    MSM(bits) + MSM(1 - bits) == MSM(ones)."""
    import numpy as np
    from bulletproofs_gadgets_amd import _native
    n = 1 << 20
    pts = gpu_ctx.points_from_uniform(synth.uniform_bytes(n, tag=b"skew20"))
    one = (1).to_bytes(32, "little")
    ones = gpu_ctx.alloc(32 * n).upload(one * n)
    q_part = gpu_ctx.msm_submit(pts, ones.ptr, n).wait_partial()
    q = _native.combine_partials(q_part)
    assert q != bytes(32)
    s = synth.scalars(1, tag=b"skew-s")
    eq = gpu_ctx.alloc(32 * n).upload(s * n)
    got = gpu_ctx.msm_submit(pts, eq.ptr, n).wait()
    assert got == gpu_ctx.msm(gpu_ctx.points_from_compressed(q), s)
    rng = np.random.default_rng(20)
    bits = rng.integers(0, 2, size=n, dtype=np.uint8)
    b = np.zeros((n, 32), dtype=np.uint8)
    b[:, 0] = bits
    nb = np.zeros((n, 32), dtype=np.uint8)
    nb[:, 0] = 1 - bits
    pb = gpu_ctx.msm_submit(pts, gpu_ctx.alloc(32 * n).upload(b.tobytes()).ptr, n).wait_partial()
    pnb = gpu_ctx.msm_submit(pts, gpu_ctx.alloc(32 * n).upload(nb.tobytes()).ptr, n).wait_partial()
    assert _native.combine_partials(pb + pnb) == q


@pytest.mark.parametrize("env", [{"BPG_MSM_LS0": "1"}, {"BPG_MSM_LS0": "5"}, {"BPG_MSM_LS0": "23", "BPG_MSM_LS1": "3"},
                                 {"BPG_MSM_C": "9", "BPG_MSM_TILE": "512"}, {"BPG_MSM_C": "12", "BPG_MSM_LS0": "64"},
                                 {"BPG_MSM_C": "6", "BPG_MSM_LS1": "2"}, {"BPG_MSM_SORT_CHUNK": "256"},
                                 {"BPG_MSM_C": "12", "BPG_MSM_SORT_CHUNK": "300"},
                                 {"BPG_MSM_C": "15"}, {"BPG_MSM_C": "16"}, {"BPG_MSM_C": "16", "BPG_MSM_LS0": "3"}])
def test_msm_planner_overrides(oracle, env):
    """every geometry the planner can be pushed into (segment size S0 not a power of two, narrow and wide
    merge passes, deep merge trees, small tiles, one and many buckets per bin, many level-2 chunks per bin) gives the oracle's bytes,
    on uniform and on skewed scalars"""
    import os
    import bulletproofs_gadgets_amd as bpg
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ctx = bpg.Context(0)              # the overrides are read when a context is created
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        n = 3000
        u = synth.uniform_bytes(n, tag=b"ovr")
        pts = ctx.points_from_uniform(u)
        arr, _ = oracle.from_uniform(u)
        # l - 1, (l - 1) / 2, (l + 1) / 2 and 2^15-aligned values: the digit that is exactly half a window at c = 16
        edge = [L - 1, (L - 1) // 2, (L + 1) // 2, 1 << 15, (1 << 15) + (1 << 31), L - (1 << 15), (1 << 252) - 1, 0x8000800080008000]
        edge_sc = _scalars_from_ints([edge[i % len(edge)] for i in range(n)])
        for sc in (synth.scalars(n, tag=b"ovr-s"), synth.small_scalars(n, 1, tag=b"ovr-b"), synth.small_scalars(n, 9, tag=b"ovr-9"),
                   synth.scalars(1, tag=b"ovr-e") * n, edge_sc):
            assert ctx.msm(pts, sc) == oracle.msm(sc, arr, n)
    finally:
        ctx.close()
