"""Host-side Python mirror of the R1CS layer of libbpg.so (include/bpg.h, second half).

Same names and argument meaning as the Rust API the reference calls
(`Prover`, `Verifier`, `LinearCombination`, `Variable`, `BulletproofGens`,
`SonnyEdwardsPointGadget`, `SonnyRistrettoPointGadget`, the gadget functions), so
that the parity tests read like the reference's own tests.  Every method is one
call through the C ABI; nothing is computed here.
"""
import ctypes
import os

from . import _native
from ._native import BpgError, check, lib

L = 2**252 + 27742317777372353535851937790883648493
KIND_COMMITTED, KIND_LEFT, KIND_RIGHT, KIND_OUTPUT, KIND_ONE = 0, 1, 2, 3, 4
ONE_VAR = KIND_ONE << 32


class VerificationError(Exception):
    """R1CSError::VerificationError"""


def _sc(x):
    return (int(x) % L).to_bytes(32, "little")


class Variable:
    __slots__ = ("code",)

    def __init__(self, code):
        self.code = int(code)

    def _lc(self):
        return LinearCombination([(self.code, 1)])

    def __add__(self, o):
        return self._lc() + o

    def __radd__(self, o):
        return LinearCombination.of(o) + self

    def __sub__(self, o):
        return self._lc() - o

    def __rsub__(self, o):
        return LinearCombination.of(o) - self

    def __neg__(self):
        return -self._lc()


class LinearCombination:
    __slots__ = ("terms",)

    def __init__(self, terms=None):
        self.terms = list(terms or [])

    @staticmethod
    def of(x):
        if isinstance(x, LinearCombination):
            return x
        if isinstance(x, Variable):
            return LinearCombination([(x.code, 1)])
        return LinearCombination([(ONE_VAR, int(x) % L)])

    def clone(self):
        return LinearCombination(self.terms)

    def __add__(self, o):
        return LinearCombination(self.terms + LinearCombination.of(o).terms)

    __radd__ = __add__

    def __sub__(self, o):
        return LinearCombination(self.terms + [(v, (-c) % L) for v, c in LinearCombination.of(o).terms])

    def __rsub__(self, o):
        return LinearCombination.of(o) - self

    def __neg__(self):
        return LinearCombination([(v, (-c) % L) for v, c in self.terms])

    def marshal(self):
        n = len(self.terms)
        vars_ = (ctypes.c_uint64 * max(n, 1))(*[v for v, _ in self.terms])
        coeffs = b"".join(_sc(c) for _, c in self.terms)
        return vars_, coeffs, n


LC = LinearCombination


class _Rng:
    """bpg_rng_fn over `data`: the external randomness of a call (the reference passes `&mut thread_rng()`,
    ref: tests/edwards_point.rs:31,79).  data=None draws from os.urandom, which is what a caller who passes nothing
    gets; a bytes object is handed out in order (seeded, reproducible runs) and a stream that runs out is an
    error, raised by check() after the call returns -- never padded with zeros."""

    def __init__(self, data):
        self.data, self.pos, self.short = data, 0, False
        self.cb = _native.RNG_FN(self._fill)

    def _fill(self, _user, out, n):
        if self.data is None:
            chunk = os.urandom(n)
        else:
            chunk = self.data[self.pos:self.pos + n]
            self.pos += n
            if len(chunk) < n:
                self.short = True
                chunk = os.urandom(n)      # the call still gets real randomness; its result is discarded below
        ctypes.memmove(out, chunk, n)

    def check(self):
        if self.short:
            raise ValueError("external randomness exhausted: %d bytes given, more were asked for" % len(self.data))


class BulletproofGens:
    """BulletproofGens::new(capacity, 1) + PedersenGens::default(), resident on the context's GPU."""

    def __init__(self, ctx, gens_capacity, party_capacity=1, shard=None):
        """shard = (rank, world, all_gather): keep only this rank's generator range resident and
        split every MSM of prove / verify by point range (bpg_gens_create_sharded).
        all_gather(mine: bytes) -> bytes of every rank's `mine`, rank-major (e.g. make_all_gather(dist))."""
        self.ctx, self.gens_capacity = ctx, gens_capacity
        h = ctypes.c_void_p()
        if shard is None:
            check(lib().bpg_gens_create(ctx._h, gens_capacity, party_capacity, ctypes.byref(h)))
        else:
            rank, world, all_gather = shard

            def fn(_user, mine, nbytes, out):
                try:
                    res = all_gather(ctypes.string_at(mine, nbytes))
                    if len(res) != nbytes * world:
                        return 1
                    ctypes.memmove(out, res, len(res))
                    return 0
                except Exception:      # noqa: BLE001  (must not unwind through the C ABI)
                    import traceback
                    traceback.print_exc()
                    return 1
            self._exchange = _native.EXCHANGE_FN(fn)      # keep the trampoline alive
            check(lib().bpg_gens_create_sharded(ctx._h, gens_capacity, party_capacity, rank, world, self._exchange, None,
                                                ctypes.byref(h)))
        self._h = h.value

    def export(self, which, start=0, n=1):
        out = ctypes.create_string_buffer(32 * n)
        check(lib().bpg_gens_export(self.ctx._h, self._h, ord(which), start, n, out))
        return out.raw

    def free(self):
        if self._h:
            lib().bpg_gens_free(self._h)
            self._h = None


class Proof:
    def __init__(self, raw):
        self.raw = raw

    def to_bytes(self):
        return self.raw


class _CS:
    def _triple(self, out):
        return Variable(out[0]), Variable(out[1]), Variable(out[2])

    def multiply(self, left, right):
        lv, lc, nl = LinearCombination.of(left).marshal()
        rv, rc, nr = LinearCombination.of(right).marshal()
        out = (ctypes.c_uint64 * 3)()
        check(lib().bpg_cs_multiply(self._cs, lv, lc, nl, rv, rc, nr, out))
        return self._triple(out)

    def allocate_multiplier(self, assignment):
        out = (ctypes.c_uint64 * 3)()
        if assignment is None:
            check(lib().bpg_cs_allocate_multiplier(self._cs, None, None, out))
        else:
            check(lib().bpg_cs_allocate_multiplier(self._cs, _sc(assignment[0]), _sc(assignment[1]), out))
        return self._triple(out)

    def constrain(self, lc):
        v, c, n = LinearCombination.of(lc).marshal()
        check(lib().bpg_cs_constrain(self._cs, v, c, n))

    def specify_randomized_constraints(self, callback):
        """RandomizableConstraintSystem::specify_randomized_constraints: `callback(rcs)` runs inside prove() /
        verify(), after the first-phase commitments; rcs offers challenge_scalar(label) beside multiply /
        allocate_multiplier / constrain.  An exception raised by the callback is re-raised by prove / verify."""
        owner = self

        def tramp(_user, rcs_ptr):
            try:
                callback(_RandomizedCS(rcs_ptr))
                return 0
            except BaseException as e:          # never unwind through the C frames
                owner._cb_error = e
                return _native.EINVAL

        fn = _native.RANDOMIZED_FN(tramp)
        self.__dict__.setdefault("_keep", []).append(fn)     # the library holds the pointer until prove / verify
        check(lib().bpg_cs_specify_randomized_constraints(self._cs, fn, None))

    def _raise_callback_error(self):
        e = self.__dict__.pop("_cb_error", None)
        if e is not None:
            raise e

    def export_terms(self):
        """[(kind, index, coeff, constraint)] for every linear-combination term captured so far (bpg_cs_export_terms)"""
        n = ctypes.c_size_t()
        lib().bpg_cs_export_terms(self._cs, None, None, None, 0, ctypes.byref(n))
        k = max(n.value, 1)
        vars_, coeffs, q = (ctypes.c_uint64 * k)(), ctypes.create_string_buffer(32 * k), (ctypes.c_uint32 * k)()
        check(lib().bpg_cs_export_terms(self._cs, vars_, coeffs, q, k, ctypes.byref(n)))
        return [(vars_[t] >> 32, vars_[t] & 0xffffffff, int.from_bytes(coeffs.raw[32 * t:32 * t + 32], "little"), q[t])
                for t in range(n.value)]


class _RandomizedCS(_CS):
    """the `rcs` handed to a randomized-constraints callback (valid during that call only)"""

    def __init__(self, ptr):
        self._cs = ptr

    def challenge_scalar(self, label):
        out = ctypes.create_string_buffer(32)
        check(lib().bpg_cs_challenge_scalar(self._cs, label, len(label), out))
        return int.from_bytes(out.raw, "little")

    def specify_randomized_constraints(self, callback):
        raise ValueError("already in the randomization phase")


class Prover(_CS):
    def __init__(self, gens, label):
        self.gens = gens
        h = ctypes.c_void_p()
        check(lib().bpg_prover_new(gens._h, label, len(label), ctypes.byref(h)))
        self._h = h.value
        self._cs = lib().bpg_prover_cs(self._h)

    def commit(self, v, v_blinding):
        V = ctypes.create_string_buffer(32)
        var = ctypes.c_uint64()
        check(lib().bpg_prover_commit(self._h, _sc(v), _sc(v_blinding), V, ctypes.byref(var)))
        return V.raw, Variable(var.value)

    def commit_batch(self, values, blindings):
        m = len(values)
        V = ctypes.create_string_buffer(32 * max(m, 1))
        vars_ = (ctypes.c_uint64 * max(m, 1))()
        check(lib().bpg_prover_commit_batch(self._h, b"".join(_sc(x) for x in values), b"".join(_sc(x) for x in blindings), m, V, vars_))
        return [V.raw[32 * i:32 * i + 32] for i in range(m)], [Variable(vars_[i]) for i in range(m)]

    def commit_batch_raw(self, values32, blindings32):
        """commit_batch over packed 32-byte little-endian scalars; returns (packed commitments,
        ctypes array of variable codes) without per-item Python work (throughput drivers)"""
        m = len(values32) // 32
        V = ctypes.create_string_buffer(32 * max(m, 1))
        vars_ = (ctypes.c_uint64 * max(m, 1))()
        check(lib().bpg_prover_commit_batch(self._h, values32, blindings32, m, V, vars_))
        return V, vars_

    def prove(self, bp_gens=None, external_rng32=None):
        """external_rng32: 32 bytes of external randomness (None: os.urandom, as `thread_rng` in the reference)"""
        if bp_gens is not None and bp_gens is not self.gens:      # Prover::prove(&bp_gens): generators named at proving time
            check(lib().bpg_prover_set_gens(self._h, bp_gens._h))
            self.gens = bp_gens
        rng = _Rng(external_rng32)
        cap = 1 << 16
        out = ctypes.create_string_buffer(cap)
        n = ctypes.c_size_t()
        rc = lib().bpg_prover_prove(self._h, rng.cb, None, out, cap, ctypes.byref(n))
        self._raise_callback_error()
        check(rc)
        rng.check()
        return Proof(ctypes.string_at(out, n.value))

    def begin_blindings(self, external_rng32):
        """One large proof: hand the 32 bytes of external randomness over now -- after the last commit, before the
        gadget calls -- and a helper thread draws the blinding vectors while this thread captures the circuit
        (bpg_prover_begin_blindings).  prove() then ignores its own external_rng32; the proof's bytes are those of
        prove(external_rng32=these bytes) without this call."""
        if len(external_rng32) != 32:
            raise ValueError("32 bytes of external randomness")
        check(lib().bpg_prover_begin_blindings(self._h, bytes(external_rng32)))

    @staticmethod
    def prove_many(provers, external_rng=None):
        """k proofs by this thread (bpg_prover_prove_many): the bytes of [p.prove() for p in provers] when the
        same external randomness stream (32 bytes per proof, in order) is used; None: os.urandom"""
        k = len(provers)
        rng = _Rng(external_rng)
        cap = 1 << 16
        out = ctypes.create_string_buffer(cap * max(k, 1))
        lens = (ctypes.c_size_t * max(k, 1))()
        hs = (ctypes.c_void_p * max(k, 1))(*[p._h for p in provers])
        rc = lib().bpg_prover_prove_many(hs, k, rng.cb, None, out, cap, lens)
        for p in provers:
            p._raise_callback_error()
        check(rc)
        rng.check()
        base = ctypes.addressof(out)           # (out.raw would copy all k * cap bytes once per proof)
        return [Proof(ctypes.string_at(base + j * cap, lens[j])) for j in range(k)]

    def stats(self):
        a, b = ctypes.c_size_t(), ctypes.c_size_t()
        check(lib().bpg_prover_stats(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def timing(self):
        t = (ctypes.c_double * 5)()
        check(lib().bpg_prover_timing(self._h, t))
        return dict(zip(("rng_ms", "commit_ms", "poly_ms", "ipp_ms", "total_ms"), t))

    KERNELS = ("k_zpow", "k_flatten", "k_poly", "k_eval", "k_ipp_rows", "k_ipp_fold")

    def kernel_timing(self):
        """{kernel: (microseconds, bytes to move, multiplications mod l)} of the last single prove() when the context was
        created with BPG_TIME_KERNELS=1 in the environment; None otherwise (bpg_prover_kernel_timing)"""
        us, by, mm = (ctypes.c_double * 6)(), (ctypes.c_double * 6)(), (ctypes.c_double * 6)()
        if lib().bpg_prover_kernel_timing(self._h, us, by, mm) != _native.OK:
            return None
        return {k: (us[i], by[i], mm[i]) for i, k in enumerate(self.KERNELS)}

    def __del__(self):
        try:
            if self._h:
                lib().bpg_prover_free(self._h)
                self._h = None
        except Exception:
            pass


class Verifier(_CS):
    def __init__(self, label):
        h = ctypes.c_void_p()
        check(lib().bpg_verifier_new(label, len(label), ctypes.byref(h)))
        self._h = h.value
        self._cs = lib().bpg_verifier_cs(self._h)

    def commit(self, V32):
        var = ctypes.c_uint64()
        check(lib().bpg_verifier_commit(self._h, V32, ctypes.byref(var)))
        return Variable(var.value)

    def commit_batch_raw(self, V):
        m = len(V) // 32
        vars_ = (ctypes.c_uint64 * max(m, 1))()
        check(lib().bpg_verifier_commit_batch(self._h, V, m, vars_))
        return vars_

    def verify(self, proof, pc_gens, bp_gens, external_rng32=None):
        """Ok(()) -> True; Err(R1CSError::VerificationError) -> raises VerificationError.
        external_rng32: 32 bytes (None: os.urandom, as `&mut thread_rng()` at ref tests/edwards_point.rs:79)"""
        raw = proof.to_bytes() if hasattr(proof, "to_bytes") else bytes(proof)
        rng = _Rng(external_rng32)
        rc = lib().bpg_verifier_verify(self._h, bp_gens._h, raw, len(raw), rng.cb, None)
        self._raise_callback_error()
        rng.check()
        if rc == _native.EVERIFY:
            raise VerificationError()
        check(rc)
        return True

    def stats(self):
        a, b = ctypes.c_size_t(), ctypes.c_size_t()
        check(lib().bpg_verifier_stats(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def __del__(self):
        try:
            if self._h:
                lib().bpg_verifier_free(self._h)
                self._h = None
        except Exception:
            pass


def verify_batch(verifiers, proofs, bp_gens, external_rng=None):
    """k proofs, one combined multiscalar multiplication (BASELINE config 5).  True iff every proof
    verifies; raises VerificationError if at least one does not.  external_rng: None (os.urandom) or
    >= 32 * k bytes, 32 per proof in order; the batching weights are derived inside the library from
    each proof's transcript and these bytes, and are never zero."""
    k = len(verifiers)
    assert len(proofs) == k
    raws = [p.to_bytes() if hasattr(p, "to_bytes") else bytes(p) for p in proofs]
    vs = (ctypes.c_void_p * max(k, 1))(*[v._h for v in verifiers])
    bufs = [ctypes.create_string_buffer(r, len(r)) for r in raws]
    ps = (ctypes.c_void_p * max(k, 1))(*[ctypes.addressof(b) for b in bufs])
    lens = (ctypes.c_size_t * max(k, 1))(*[len(r) for r in raws])
    rng = _Rng(external_rng)
    rc = lib().bpg_verifier_verify_batch(vs, ps, lens, k, bp_gens._h, rng.cb, None)
    for v in verifiers:
        v._raise_callback_error()
    rng.check()
    if rc == _native.EVERIFY:
        raise VerificationError()
    check(rc)
    return True


def make_all_gather(dist, device=None):
    """all_gather(mine) over torch.distributed for BulletproofGens(shard=...).  With device set
    (e.g. "cuda") the 128-byte partial sums travel through device tensors, i.e. over RCCL / xGMI
    with the "nccl" backend; device=None uses CPU tensors (gloo)."""
    import torch

    def all_gather(mine):
        world = dist.get_world_size()
        src = torch.frombuffer(bytearray(mine), dtype=torch.uint8)
        if device is not None:
            src = src.to(device)
        out = torch.empty(len(mine) * world, dtype=torch.uint8, device=src.device)
        dist.all_gather_into_tensor(out, src)
        return out.cpu().numpy().tobytes()
    return all_gather


def _xyzt(p):
    return _sc(p.X) + _sc(p.Y) + _sc(p.Z) + _sc(p.T)


class _PointGadget:
    def __init__(self, handle):
        self._h = handle

    @classmethod
    def _new(cls, fn, *args):
        h = ctypes.c_void_p()
        check(fn(*args, ctypes.byref(h)))
        return cls(h.value)

    def __del__(self):
        try:
            if self._h:
                lib().bpg_point_free(self._h)
                self._h = None
        except Exception:
            pass


class SonnyEdwardsPointGadget(_PointGadget):
    def __init__(self, X=None, Y=None, Z=None, T=None):
        if Y is None:
            super().__init__(X)
            return
        import types
        h = ctypes.c_void_p()
        check(lib().bpg_point_from_coords(_xyzt(types.SimpleNamespace(X=X, Y=Y, Z=Z, T=T)), ctypes.byref(h)))
        super().__init__(h.value)

    @classmethod
    def from_point(cls, p):
        return cls._new(lib().bpg_point_from_coords, _xyzt(p))

    @classmethod
    def from_vars(cls, vars_):
        arr = (ctypes.c_uint64 * 4)(*[v.code for v in vars_])
        return cls._new(lib().bpg_point_from_vars, arr)

    def add(self, other, cs):
        return self._new(lib().bpg_edwards_add, cs._cs, self._h, other._h)

    def double(self, cs):
        return self._new(lib().bpg_edwards_double, cs._cs, self._h)

    def equal(self, other, cs):
        check(lib().bpg_edwards_equal(cs._cs, self._h, other._h))

    def satisfy_curve_eq(self, cs):
        check(lib().bpg_edwards_satisfy_curve_eq(cs._cs, self._h))

    def conditionally_select(self, bit, cs):
        v, c, n = LinearCombination.of(bit).marshal()
        return self._new(lib().bpg_edwards_conditionally_select, cs._cs, self._h, v, c, n)

    @classmethod
    def scalar_mul(cls, point, sk, cs):
        arr = sk if isinstance(sk, ctypes.Array) else (ctypes.c_uint64 * max(len(sk), 1))(*[v.code for v in sk])
        return cls._new(lib().bpg_edwards_scalar_mul, cs._cs, point._h, arr, len(sk))

    @classmethod
    def prover_commit_to_sonny_edwards_point(cls, prover, p, rng):
        bl = b"".join(_sc(rng.scalar()) for _ in range(4))
        V = ctypes.create_string_buffer(128)
        h = ctypes.c_void_p()
        check(lib().bpg_edwards_prover_commit(prover._h, _xyzt(p), bl, ctypes.byref(h), V))
        return cls(h.value), [V.raw[32 * i:32 * i + 32] for i in range(4)]

    @classmethod
    def verifier_commit_to_sonny_edwards_point(cls, verifier, commitments):
        return cls._new(lib().bpg_edwards_verifier_commit, verifier._h, b"".join(commitments), len(commitments))


class SonnyRistrettoPointGadget(_PointGadget):
    @classmethod
    def from_point(cls, point, cs):
        return cls._new(lib().bpg_ristretto_from_point, cs._cs, _xyzt(point.p))

    @classmethod
    def from_lcs(cls, gadget, n_lcs, cs):
        return cls._new(lib().bpg_ristretto_from_lcs, cs._cs, gadget._h, n_lcs)

    def add(self, cs, other):
        return self._new(lib().bpg_ristretto_add, cs._cs, self._h, other._h)

    def double(self, cs):
        return self._new(lib().bpg_ristretto_double, cs._cs, self._h)

    def equals(self, cs, other):
        check(lib().bpg_ristretto_equals(cs._cs, self._h, other._h))

    def conditionally_select(self, bit, cs):
        v, c, n = LinearCombination.of(bit).marshal()
        return self._new(lib().bpg_ristretto_conditionally_select, cs._cs, self._h, v, c, n)


def binary_constrain_gadget(cs, bit):
    check(lib().bpg_gadget_binary_constrain(cs._cs, bit.code))


def nonzero_gadget(var, var_assignment, cs):
    """raises RuntimeError where the reference panics (zero assignment)"""
    v, c, n = LinearCombination.of(var).marshal()
    rc = lib().bpg_gadget_nonzero(cs._cs, v, c, n, None if var_assignment is None else _sc(var_assignment))
    if rc == _native.EGADGET:
        raise RuntimeError((lib().bpg_last_error() or b"").decode())
    check(rc)


def sk_knowledge_gadget(cs, basep, pk, sk):
    arr = (ctypes.c_uint64 * max(len(sk), 1))(*[v.code for v in sk])
    check(lib().bpg_sk_knowledge_gadget(cs._cs, basep._h, pk._h, arr, len(sk)))


def transcript_kat(label, msg_label, msg, challenge_label, n):
    out = ctypes.create_string_buffer(n)
    check(lib().bpg_transcript_kat(label, len(label), msg_label, msg, len(msg), challenge_label, out, n))
    return out.raw


__all__ = ["BpgError", "BulletproofGens", "LinearCombination", "LC", "Variable", "Prover", "Verifier", "Proof",
           "VerificationError", "verify_batch", "make_all_gather", "SonnyEdwardsPointGadget", "SonnyRistrettoPointGadget", "binary_constrain_gadget",
           "nonzero_gadget", "sk_knowledge_gadget", "transcript_kat"]
