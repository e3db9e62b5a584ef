"""Deterministic synthetic inputs: uniform Ristretto points
and uniform scalars mod l from a SHAKE256 stream.  Pure hashlib / integers, no
GPU and no oracle, so tests, bench.py and the oracle all see the same bytes."""
import hashlib

L = 2**252 + 27742317777372353535851937790883648493
SEED = b"bpg-bench-v1"


def uniform_bytes(n, seed=SEED, tag=b"P"):
    """n x 64 bytes; point i = RistrettoPoint::from_uniform_bytes(chunk i)."""
    return hashlib.shake_256(seed + tag).digest(64 * n)


def scalars(n, seed=SEED, tag=b"s"):
    """n x 32 bytes, each Scalar::from_bytes_mod_order_wide of a 64-byte chunk."""
    raw = hashlib.shake_256(seed + tag).digest(64 * n)
    out = bytearray(32 * n)
    for i in range(n):
        out[32 * i:32 * i + 32] = (int.from_bytes(raw[64 * i:64 * i + 64], "little") % L).to_bytes(32, "little")
    return bytes(out)


def small_scalars(n, bits, seed=SEED, tag=b"b"):
    """n scalars below 2^bits (bit / small-value witnesses as in the gadget circuits)."""
    raw = hashlib.shake_256(seed + tag).digest(8 * n)
    out = bytearray(32 * n)
    mask = (1 << bits) - 1
    for i in range(n):
        out[32 * i:32 * i + 8] = (int.from_bytes(raw[8 * i:8 * i + 8], "little") & mask).to_bytes(8, "little")
    return bytes(out)
