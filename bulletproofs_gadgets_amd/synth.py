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


# ---- witnesses for the proof configs --------------------------
# zerocaf's "Sonny" curve: -x^2 + y^2 = 1 + d x^2 y^2 over F_l, d = -86650/86649.  Plain
# integers; only used to make a statement (base, bits, base*k) for the gadget circuits.
SONNY_D = (-86650 * pow(86649, -1, L)) % L


class SonnyPoint:
    __slots__ = ("X", "Y", "Z", "T")

    def __init__(self, X, Y, Z, T):
        self.X, self.Y, self.Z, self.T = X % L, Y % L, Z % L, T % L

    def __add__(self, q):
        # add-2008-hwcd with a = -1
        A, B = self.X * q.X % L, self.Y * q.Y % L
        C, D = SONNY_D * self.T * q.T % L, self.Z * q.Z % L
        E = ((self.X + self.Y) * (q.X + q.Y) - A - B) % L
        F, G, H = D - C, D + C, B + A
        return SonnyPoint(E * F, G * H, F * G, E * H)

    def mul_bits(self, bits_lsb_first):
        acc = SonnyPoint(0, 1, 1, 0)
        for b in reversed(bits_lsb_first):
            acc = acc + acc
            if b:
                acc = acc + self
        return acc


def _sqrt_mod_l(a):
    """square root mod l (l = 5 mod 8), Atkin; raises if a is a non-residue"""
    a %= L
    r = pow(a, (L + 3) // 8, L)
    if r * r % L != a:
        r = r * pow(2, (L - 1) // 4, L) % L
    if r * r % L != a:
        raise ValueError("non-residue")
    return r


def sonny_base():
    """8 * (the point with y = 8/9 and even x): a point of the prime-order subgroup"""
    y = 8 * pow(9, -1, L) % L
    x = _sqrt_mod_l((y * y - 1) * pow(SONNY_D * y * y + 1, -1, L))
    if x & 1:
        x = L - x
    p = SonnyPoint(x, y, 1, x * y)
    return p.mul_bits([0, 0, 0, 1])


def scalar_mul_statement(nbits, seed=SEED):
    """cfg3 / cfg4: (base, bits[nbits] LSB first, base * k); nbits = 2427 -> 65 533 multipliers"""
    raw = hashlib.shake_256(seed + b"bits").digest((nbits + 7) // 8)
    bits = [(raw[i >> 3] >> (i & 7)) & 1 for i in range(nbits)]
    base = sonny_base()
    return base, bits, base.mul_bits(bits)
