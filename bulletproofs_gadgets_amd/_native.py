"""ctypes binding of libbpg.so (include/bpg.h).

The HIP library is the product; there is no Python or CPU fallback.  Importing
this module never touches the GPU; `Context()` does, and raises if the
extension or the device is missing.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libbpg.so")

OK = 0
EINVAL, EDEVICE, EFORMAT, EVERIFY, EGENS, EMISSING, EGADGET, ENOMEM = -1, -2, -3, -4, -5, -6, -7, -8
_ERR_NAMES = {
    EINVAL: "BPG_EINVAL", EDEVICE: "BPG_EDEVICE", EFORMAT: "BPG_EFORMAT", EVERIFY: "BPG_EVERIFY",
    EGENS: "BPG_EGENS", EMISSING: "BPG_EMISSING", EGADGET: "BPG_EGADGET", ENOMEM: "BPG_ENOMEM",
}


class BpgError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        super().__init__("%s (%d)%s" % (_ERR_NAMES.get(code, "BPG_E?"), code, ": " + detail if detail else ""))


_lib = None
# bpg_rng_fn: void (*)(void *user, uint8_t *out, size_t n)
RNG_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_size_t)
RANDOMIZED_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)
EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)


def lib():
    """Load libbpg.so (built in-tree by build.sh / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            "bulletproofs_gadgets_amd: %s is missing. Build it with ./build.sh (hipcc, gfx950). "
            "There is no CPU fallback." % _LIB_PATH)
    L = ctypes.CDLL(_LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    vp, sz, cp, ip = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)
    pvp = ctypes.POINTER(ctypes.c_void_p)
    fp = ctypes.POINTER(ctypes.c_float)
    psz = ctypes.POINTER(ctypes.c_size_t)
    sigs = {
        "bpg_init": (ctypes.c_int, [ctypes.c_int, pvp]),
        "bpg_shutdown": (None, [vp]),
        "bpg_last_error": (cp, []),
        "bpg_version": (cp, []),
        "bpg_device_alloc": (ctypes.c_int, [vp, sz, pvp]),
        "bpg_device_free": (ctypes.c_int, [vp, vp]),
        "bpg_copy_to_device": (ctypes.c_int, [vp, vp, cp, sz]),
        "bpg_copy_to_host": (ctypes.c_int, [vp, vp, vp, sz]),
        "bpg_synchronize": (ctypes.c_int, [vp]),
        "bpg_points_from_compressed": (ctypes.c_int, [vp, cp, sz, pvp]),
        "bpg_points_from_uniform": (ctypes.c_int, [vp, cp, sz, pvp]),
        "bpg_points_compress": (ctypes.c_int, [vp, vp, sz, sz, vp]),
        "bpg_points_len": (sz, [vp]),
        "bpg_points_free": (None, [vp]),
        "bpg_msm": (ctypes.c_int, [vp, vp, cp, sz, vp]),
        "bpg_msm_submit": (ctypes.c_int, [vp, vp, vp, sz, sz, pvp]),
        "bpg_msm_submit_segments": (ctypes.c_int, [vp, vp, psz, psz, sz, vp, sz, pvp]),
        "bpg_msm_wait": (ctypes.c_int, [vp, vp]),
        "bpg_msm_wait_partial": (ctypes.c_int, [vp, vp]),
        "bpg_msm_combine_partials": (ctypes.c_int, [cp, sz, vp]),
        "bpg_msm_plan": (ctypes.c_int, [vp, sz, sz, ip, ip, ip, ip]),
        "bpg_msm_last_kernel_us": (ctypes.c_int, [vp, fp, fp]),
        "bpg_debug_fail_alloc": (None, [ctypes.c_int]),
        "bpg_msm_noalloc": (ctypes.c_int, [vp, vp, vp, sz, vp, vp]),
    }
    u64p, dp = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_double)
    ci = ctypes.c_int
    sigs.update({
        "bpg_gens_create": (ci, [vp, sz, sz, pvp]),
        "bpg_gens_create_sharded": (ci, [vp, sz, sz, ci, ci, EXCHANGE_FN, vp, pvp]),
        "bpg_gens_free": (None, [vp]),
        "bpg_gens_capacity": (sz, [vp]),
        "bpg_gens_export": (ci, [vp, vp, ci, sz, sz, vp]),
        "bpg_prover_new": (ci, [vp, cp, sz, pvp]),
        "bpg_prover_free": (None, [vp]),
        "bpg_prover_cs": (vp, [vp]),
        "bpg_prover_commit": (ci, [vp, cp, cp, vp, u64p]),
        "bpg_prover_commit_batch": (ci, [vp, cp, cp, sz, vp, u64p]),
        "bpg_prover_prove": (ci, [vp, RNG_FN, vp, vp, sz, psz]),
        "bpg_prover_prove_many": (ci, [vp, sz, RNG_FN, vp, vp, sz, psz]),
        "bpg_prover_begin_blindings": (ci, [vp, cp]),
        "bpg_prover_stats": (ci, [vp, psz, psz]),
        "bpg_prover_timing": (ci, [vp, dp]),
        "bpg_prover_kernel_timing": (ci, [vp, dp, dp, dp]),
        "bpg_verifier_new": (ci, [cp, sz, pvp]),
        "bpg_verifier_free": (None, [vp]),
        "bpg_verifier_cs": (vp, [vp]),
        "bpg_verifier_commit": (ci, [vp, cp, u64p]),
        "bpg_verifier_commit_batch": (ci, [vp, cp, sz, u64p]),
        "bpg_verifier_verify_batch": (ci, [vp, vp, vp, sz, vp, RNG_FN, vp]),
        "bpg_verifier_verify": (ci, [vp, vp, cp, sz, RNG_FN, vp]),
        "bpg_verifier_stats": (ci, [vp, psz, psz]),
        "bpg_cs_multiply": (ci, [vp, u64p, cp, sz, u64p, cp, sz, u64p]),
        "bpg_cs_allocate_multiplier": (ci, [vp, cp, cp, u64p]),
        "bpg_cs_constrain": (ci, [vp, u64p, cp, sz]),
        "bpg_prover_set_gens": (ci, [vp, vp]),
        "bpg_cs_specify_randomized_constraints": (ci, [vp, RANDOMIZED_FN, vp]),
        "bpg_cs_challenge_scalar": (ci, [vp, cp, sz, cp]),
        "bpg_flat_cache_stats": (ci, [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
        "bpg_cs_export_terms": (ci, [vp, u64p, vp, ctypes.POINTER(ctypes.c_uint32), sz, psz]),
        "bpg_gadget_binary_constrain": (ci, [vp, ctypes.c_uint64]),
        "bpg_gadget_nonzero": (ci, [vp, u64p, cp, sz, cp]),
        "bpg_point_from_coords": (ci, [cp, pvp]),
        "bpg_point_from_vars": (ci, [u64p, pvp]),
        "bpg_point_free": (None, [vp]),
        "bpg_edwards_add": (ci, [vp, vp, vp, pvp]),
        "bpg_edwards_double": (ci, [vp, vp, pvp]),
        "bpg_edwards_equal": (ci, [vp, vp, vp]),
        "bpg_edwards_satisfy_curve_eq": (ci, [vp, vp]),
        "bpg_edwards_conditionally_select": (ci, [vp, vp, u64p, cp, sz, pvp]),
        "bpg_edwards_scalar_mul": (ci, [vp, vp, u64p, sz, pvp]),
        "bpg_edwards_prover_commit": (ci, [vp, cp, cp, pvp, vp]),
        "bpg_edwards_verifier_commit": (ci, [vp, cp, sz, pvp]),
        "bpg_ristretto_from_point": (ci, [vp, cp, pvp]),
        "bpg_ristretto_from_lcs": (ci, [vp, vp, sz, pvp]),
        "bpg_ristretto_add": (ci, [vp, vp, vp, pvp]),
        "bpg_ristretto_double": (ci, [vp, vp, pvp]),
        "bpg_ristretto_equals": (ci, [vp, vp, vp]),
        "bpg_ristretto_conditionally_select": (ci, [vp, vp, u64p, cp, sz, pvp]),
        "bpg_sk_knowledge_gadget": (ci, [vp, vp, vp, u64p, sz]),
        "bpg_transcript_kat": (ci, [cp, sz, cp, cp, sz, cp, vp, sz]),
    })
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)        # AttributeError here == header / library mismatch
        fn.restype = res
        fn.argtypes = args
    L._bpg_sigs = sigs
    _lib = L
    return L


def exported_symbols():
    """Names declared in include/bpg.h that this binding expects the library to export."""
    return sorted(lib()._bpg_sigs.keys())


def check(rc):
    if rc != OK:
        raise BpgError(rc, (lib().bpg_last_error() or b"").decode(errors="replace"))


class Points:
    """Resident table of points in HBM (affine Niels form)."""

    def __init__(self, ctx, handle):
        self.ctx, self._h = ctx, handle

    def __len__(self):
        return lib().bpg_points_len(self._h)

    def compress(self, start=0, n=None):
        n = len(self) - start if n is None else n
        out = ctypes.create_string_buffer(32 * n if n else 1)
        check(lib().bpg_points_compress(self.ctx._h, self._h, start, n, out))
        return out.raw[: 32 * n]

    def free(self):
        if self._h:
            lib().bpg_points_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceBuffer:
    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, nbytes
        p = ctypes.c_void_p()
        check(lib().bpg_device_alloc(ctx._h, nbytes, ctypes.byref(p)))
        self.ptr = p.value

    def upload(self, data):
        assert len(data) <= self.nbytes
        check(lib().bpg_copy_to_device(self.ctx._h, self.ptr, data, len(data)))
        return self

    def download(self, nbytes=None):
        nbytes = self.nbytes if nbytes is None else nbytes
        out = ctypes.create_string_buffer(nbytes if nbytes else 1)
        check(lib().bpg_copy_to_host(self.ctx._h, out, self.ptr, nbytes))
        return out.raw[:nbytes]

    def free(self):
        if self.ptr:
            lib().bpg_device_free(self.ctx._h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class MsmJob:
    def __init__(self, ctx, handle, n_targets):
        self.ctx, self._h, self.n_targets = ctx, handle, n_targets

    def wait(self):
        out = ctypes.create_string_buffer(32 * self.n_targets)
        check(lib().bpg_msm_wait(self._h, out))
        return out.raw

    def wait_partial(self):
        """n_targets x 128 B un-normalised extended points (for point-range sharding)."""
        out = ctypes.create_string_buffer(128 * self.n_targets)
        check(lib().bpg_msm_wait_partial(self._h, out))
        return out.raw


def combine_partials(partials128):
    """compress(sum of 128-byte partial sums), e.g. one per rank after an all-gather."""
    assert len(partials128) % 128 == 0
    out = ctypes.create_string_buffer(32)
    check(lib().bpg_msm_combine_partials(partials128, len(partials128) // 128, out))
    return out.raw


class Context:
    """One per process / GPU."""

    def __init__(self, device=0):
        h = ctypes.c_void_p()
        check(lib().bpg_init(device, ctypes.byref(h)))
        self._h = h.value
        self.device = device

    def close(self):
        if self._h:
            lib().bpg_shutdown(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def synchronize(self):
        check(lib().bpg_synchronize(self._h))

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def points_from_compressed(self, data):
        assert len(data) % 32 == 0
        h = ctypes.c_void_p()
        check(lib().bpg_points_from_compressed(self._h, data, len(data) // 32, ctypes.byref(h)))
        return Points(self, h.value)

    def points_from_uniform(self, data):
        assert len(data) % 64 == 0
        h = ctypes.c_void_p()
        check(lib().bpg_points_from_uniform(self._h, data, len(data) // 64, ctypes.byref(h)))
        return Points(self, h.value)

    def msm(self, points, scalars):
        """compress(sum_i scalars[i] * points[i]); scalars = n x 32 bytes."""
        assert len(scalars) % 32 == 0
        out = ctypes.create_string_buffer(32)
        check(lib().bpg_msm(self._h, points._h, scalars, len(scalars) // 32, out))
        return out.raw

    def msm_submit(self, points, d_scalars_ptr, n, n_targets=1, segments=None):
        h = ctypes.c_void_p()
        if segments is None:
            check(lib().bpg_msm_submit(self._h, points._h, d_scalars_ptr, n, n_targets, ctypes.byref(h)))
        else:
            k = len(segments)
            starts = (ctypes.c_size_t * k)(*[s for s, _ in segments])
            lens = (ctypes.c_size_t * k)(*[l for _, l in segments])
            assert sum(l for _, l in segments) == n
            check(lib().bpg_msm_submit_segments(self._h, points._h, starts, lens, k, d_scalars_ptr, n_targets,
                                                ctypes.byref(h)))
        return MsmJob(self, h.value, n_targets)

    def msm_noalloc(self, points, d_scalars_ptr, n, d_out_ptr, stream=None):
        """bpg_msm_noalloc: enqueue on `stream` (a hipStream_t as an int, None = null stream); result lands in device memory"""
        check(lib().bpg_msm_noalloc(self._h, points._h, d_scalars_ptr, n, d_out_ptr, stream))

    def msm_plan(self, n, n_targets=1):
        c, w, nb, p = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(lib().bpg_msm_plan(self._h, n, n_targets, ctypes.byref(c), ctypes.byref(w), ctypes.byref(nb), ctypes.byref(p)))
        return {"window_bits": c.value, "windows": w.value, "buckets": nb.value, "threads_per_bucket": p.value}

    def flat_cache_stats(self):
        """(hits, misses) of the circuit-structure cache since this context was created"""
        h, m = ctypes.c_uint64(), ctypes.c_uint64()
        check(lib().bpg_flat_cache_stats(self._h, ctypes.byref(h), ctypes.byref(m)))
        return h.value, m.value

    def msm_last_kernel_us(self):
        a, t = ctypes.c_float(), ctypes.c_float()
        check(lib().bpg_msm_last_kernel_us(self._h, ctypes.byref(a), ctypes.byref(t)))
        return a.value, t.value
