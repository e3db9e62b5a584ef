"""GPU: bpg_msm_noalloc, the entry point that allocates nothing (static device workspace, caller's stream, result in
device memory), against bpg_msm_submit / bpg_msm_wait on the same inputs -- byte equality (integer work).  Two calls
in a row on the same GPU share the workspace: the second, smaller one must not see anything of the first."""
import ctypes

import pytest

from bulletproofs_gadgets_amd import _native, synth

pytestmark = pytest.mark.gpu


def _both(ctx, pts, sc, n, d_sc, d_out):
    d_sc.upload(sc[:32 * n] if n else b"\0" * 32)
    ctx.msm_noalloc(pts, d_sc.ptr, n, d_out.ptr)
    ctx.synchronize()
    got = d_out.download(32)
    want = ctx.msm_submit(pts, d_sc.ptr, n).wait()
    return got, want


def test_noalloc_matches_submit_wait(gpu_ctx):
    import bulletproofs_gadgets_amd as bpg
    nmax = 16384
    pts = gpu_ctx.points_from_uniform(synth.uniform_bytes(nmax, tag=b"na"))
    sc = synth.scalars(nmax, tag=b"na-s")
    d_sc, d_out = gpu_ctx.alloc(32 * nmax), gpu_ctx.alloc(32)
    # two calls in a row, the second smaller (the shared workspace keeps the first call's contents)
    g1, w1 = _both(gpu_ctx, pts, sc, nmax, d_sc, d_out)
    assert g1 == w1
    g2, w2 = _both(gpu_ctx, pts, sc, 1000, d_sc, d_out)
    assert g2 == w2 and g2 != g1
    # sizes around the planner's window changes, tiny inputs, the empty sum, and back up to the maximum
    for n in (0, 1, 2, 3, 63, 64, 65, 189, 190, 500, 799, 800, 1024, 4096, 5000, 9999, nmax, 17, nmax):
        g, w = _both(gpu_ctx, pts, sc, n, d_sc, d_out)
        assert g == w, n
    # skewed scalars (bit witnesses: everything lands in one bucket per window) and edge scalars
    L = 2**252 + 27742317777372353535851937790883648493
    for vals in ([1] * 3000, [0] * 100 + [1] * 100, [L - 1] * 2000, [i % 2 for i in range(4097)], [2**252] * 777):
        b = b"".join(int(v).to_bytes(32, "little") for v in vals)
        g, w = _both(gpu_ctx, pts, b, len(vals), d_sc, d_out)
        assert g == w
    # the result agrees with the host-scalar entry point too
    assert _both(gpu_ctx, pts, sc, 2500, d_sc, d_out)[0] == gpu_ctx.msm(pts, sc[:32 * 2500])
    # beyond the maximum: a clear error, nothing enqueued
    big = gpu_ctx.points_from_uniform(synth.uniform_bytes(nmax + 1, tag=b"na2"))
    d_big = gpu_ctx.alloc(32 * (nmax + 1))
    with pytest.raises(bpg.BpgError) as e:
        gpu_ctx.msm_noalloc(big, d_big.ptr, nmax + 1, d_out.ptr)
    assert e.value.code == _native.EINVAL and b"NOALLOC_MAX" in _native.lib().bpg_last_error()
    g, w = _both(gpu_ctx, pts, sc, 300, d_sc, d_out)          # and the entry point still works afterwards
    assert g == w


def test_noalloc_on_a_caller_stream_and_without_allocating(gpu_ctx):
    """on a non-default stream created by the caller, many calls back to back without host synchronisation in
    between (the stream orders them: each reuses the workspace after the previous one is done), while the HIP
    allocator is watched: free device memory does not move across the calls"""
    hip = ctypes.CDLL("libamdhip64.so")
    stream = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(stream)) == 0
    try:
        n = 6000
        pts = gpu_ctx.points_from_uniform(synth.uniform_bytes(n, tag=b"na3"))
        sc = synth.scalars(n, tag=b"na3-s")
        sizes = [6000, 100, 3000, 1, 4500, 64]
        d_sc = gpu_ctx.alloc(32 * n).upload(sc)
        outs = [gpu_ctx.alloc(32) for _ in sizes]
        want = [gpu_ctx.msm_submit(pts, d_sc.ptr, k).wait() for k in sizes]
        gpu_ctx.synchronize()
        free0, free1, total = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(free0), ctypes.byref(total)) == 0
        for k, o in zip(sizes, outs):
            gpu_ctx.msm_noalloc(pts, d_sc.ptr, k, o.ptr, stream)
        assert hip.hipStreamSynchronize(stream) == 0
        assert hip.hipMemGetInfo(ctypes.byref(free1), ctypes.byref(total)) == 0
        assert free1.value == free0.value
        assert [o.download(32) for o in outs] == want
    finally:
        hip.hipStreamDestroy(stream)
