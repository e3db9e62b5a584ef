"""oracle/binding.py -- TEST INFRASTRUCTURE (see oracle/oracle.h).

ctypes access to oracle/liboracle.so for tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product never imports this module.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liboracle.so")

L = 2**252 + 27742317777372353535851937790883648493

SCHED_VARTIME, SCHED_STRAUS_CT, SCHED_STRAUS_VT, SCHED_PIPPENGER, SCHED_NAIVE = 0, 1, 2, 3, 4


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        self.GE = lib.oracle_sizeof_ge()

    # ---- points
    def ge_array(self, n):
        return ctypes.create_string_buffer(self.GE * max(n, 1))

    def from_uniform(self, uniform64):
        """n x 64 bytes -> ge array (RistrettoPoint::from_uniform_bytes each)."""
        n = len(uniform64) // 64
        arr = self.ge_array(n)
        for i in range(n):
            self.lib.ristretto_from_uniform_bytes(ctypes.byref(arr, self.GE * i), uniform64[64 * i:64 * i + 64])
        return arr, n

    def decode(self, points32):
        n = len(points32) // 32
        arr = self.ge_array(n)
        rc = self.lib.oracle_decode_many(arr, points32, n)
        return (arr, n) if rc == 0 else (None, rc)

    def encode(self, arr, n):
        out = ctypes.create_string_buffer(32 * max(n, 1))
        self.lib.oracle_encode_many(out, arr, n)
        return out.raw[:32 * n]

    def msm(self, scalars32, arr, n, schedule=SCHED_VARTIME):
        out = ctypes.create_string_buffer(32)
        self.lib.oracle_msm_ge(out, scalars32, arr, ctypes.c_size_t(n), schedule)
        return out.raw

    def msm_partial(self, scalars32, arr, n, schedule=SCHED_VARTIME):
        """128-byte un-normalised extended result (the sharding wire format)."""
        out = ctypes.create_string_buffer(128)
        self.lib.oracle_msm_ge_ext(out, scalars32, arr, ctypes.c_size_t(n), schedule)
        return out.raw

    def msm_compressed(self, scalars32, points32, schedule=SCHED_VARTIME):
        out = ctypes.create_string_buffer(32)
        rc = self.lib.oracle_msm_compressed(out, scalars32, points32, ctypes.c_size_t(len(points32) // 32), schedule)
        return out.raw if rc == 0 else None

    def gens(self, n):
        """(G, H) ge arrays of the party-0 BulletproofGens share."""
        G, H = self.ge_array(n), self.ge_array(n)
        self.lib.bulletproof_gens(G, H, ctypes.c_size_t(n))
        return G, H

    def pedersen(self):
        B, Bb = self.ge_array(1), self.ge_array(1)
        self.lib.pedersen_gens_default(B, Bb)
        return B, Bb

    def concat(self, *arrs_and_counts):
        total = sum(n for _, n in arrs_and_counts)
        out = self.ge_array(total)
        off = 0
        for arr, n in arrs_and_counts:
            ctypes.memmove(ctypes.byref(out, off), arr, self.GE * n)
            off += self.GE * n
        return out, total

    def slice(self, arr, start, n):
        out = self.ge_array(n)
        ctypes.memmove(out, ctypes.byref(arr, self.GE * start), self.GE * n)
        return out

    def fold_points(self, arr, half, s_lo, s_hi, shared):
        out = self.ge_array(half)
        self.lib.oracle_fold_points(out, arr, ctypes.c_size_t(half), s_lo, s_hi, ctypes.c_size_t(0 if shared else 1))
        return out


def load():
    if not os.path.exists(_PATH):
        raise ImportError("oracle/liboracle.so missing: run `make -C oracle` (or ./build.sh)")
    lib = ctypes.CDLL(_PATH)
    lib.oracle_sizeof_ge.restype = ctypes.c_size_t
    lib.oracle_decode_many.restype = ctypes.c_int
    lib.oracle_msm_compressed.restype = ctypes.c_int
    lib.ristretto_decode.restype = ctypes.c_int
    return Oracle(lib)
