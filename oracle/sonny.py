"""oracle/sonny.py -- TEST INFRASTRUCTURE.

The "Sonny" (formerly Doppio) curve of dusk-zerocaf, which the reference uses
only to produce WITNESS VALUES for its gadgets (ref: src/gadgets/point/edwards_point.rs:6,
ristretto_point.rs:6-8, tests/edwards_point.rs:14-15).  zerocaf is an un-vendored git
dependency (ref: Cargo.toml:10, branch master, no lockfile); what follows restates its
published parameters (UPSTREAM-RECALL, zerocaf README):

    base field   F_l,  l = 2^252 + 27742317777372353535851937790883648493
                 (the ristretto255 scalar field -- which is why a Sonny coordinate
                 is a bulletproofs Scalar, ref: edwards_point.rs:21)
    curve        -x^2 + y^2 = 1 + d x^2 y^2,  a = -1,  d = -86650/86649
                 (checked here: with this d the point y = 8/9 exists and has order r;
                 with the inverted fraction the curve is the quadratic twist)
    group order  8 * r,  r = 2^249 - 15145038707218910765482344729778085401
    basepoint    8 * (y = 8/9, x even): a fixed point of prime order r (asserted below).
                 zerocaf's own BASEPOINT constant cannot be reproduced from memory; the
                 gadgets accept any curve point, so this only fixes the test witnesses.

The circuits only need (X, Y, Z, T) values that satisfy the gadget equations, so the
exact choice of basepoint does not affect any parity claim; it is recorded here so
the fixtures are reproducible.
"""
L = 2**252 + 27742317777372353535851937790883648493
A = L - 1
D = (-86650 * pow(86649, -1, L)) % L
R_ORDER = 2**249 - 15145038707218910765482344729778085401


def _sqrt(a):
    """square root mod l (l = 5 mod 8), or None"""
    a %= L
    if a == 0:
        return 0
    if pow(a, (L - 1) // 2, L) != 1:
        return None
    x = pow(a, (L + 3) // 8, L)
    if x * x % L != a:
        x = x * pow(2, (L - 1) // 4, L) % L
    assert x * x % L == a
    return x


class EdwardsPoint:
    """zerocaf::edwards::EdwardsPoint {X, Y, Z, T} (extended twisted Edwards coordinates)."""
    __slots__ = ("X", "Y", "Z", "T")

    def __init__(self, X, Y, Z, T):
        self.X, self.Y, self.Z, self.T = X % L, Y % L, Z % L, T % L

    @staticmethod
    def identity():
        return EdwardsPoint(0, 1, 1, 0)

    @staticmethod
    def from_y(y, x_even=True):
        x2 = (y * y - 1) * pow(D * y * y + 1, -1, L) % L    # -x^2 + y^2 = 1 + d x^2 y^2
        x = _sqrt(x2)
        if x is None:
            return None
        if (x & 1) == (1 if x_even else 0):
            x = L - x
        return EdwardsPoint(x, y, 1, x * y)

    def __add__(self, o):
        # add-2008-hwcd with a = -1 (complete: d is a non-square), the very formula the
        # gadgets arithmetise (ref: edwards_point.rs:33-42): A = X1X2, B = Y1Y2, C = d T1T2,
        # D = Z1Z2, E = (X1+Y1)(X2+Y2) - A - B, F = D - C, G = D + C, H = B + A.
        a = self.X * o.X % L
        b = self.Y * o.Y % L
        c = D * self.T % L * o.T % L
        d = self.Z * o.Z % L
        e = ((self.X + self.Y) * (o.X + o.Y) - a - b) % L
        f, g, h = d - c, d + c, b + a
        return EdwardsPoint(e * f, g * h, f * g, e * h)

    def double(self):
        # The Ristretto gadget computes 8P by three circuit additions P+P
        # (ref: ristretto_point.rs:57-59,177-180) and then needs 1 / (8P).X for exactly that
        # coordinate vector (ref: ristretto_point.rs:64-69).  A dedicated doubling formula
        # (dbl-2008-hwcd) returns the same point with every coordinate negated, for which the
        # reference's own positive test could not verify; so doubling is add(self, self) here,
        # and the product's witness code does the same.
        return self + self

    def __neg__(self):
        return EdwardsPoint(-self.X, self.Y, self.Z, -self.T)

    def __mul__(self, k):
        acc, base = EdwardsPoint.identity(), self
        while k:
            if k & 1:
                acc = acc + base
            base = base.double()
            k >>= 1
        return acc

    def affine(self):
        zi = pow(self.Z, -1, L)
        return self.X * zi % L, self.Y * zi % L

    def __eq__(self, o):
        return (self.X * o.Z - o.X * self.Z) % L == 0 and (self.Y * o.Z - o.Y * self.Z) % L == 0

    def on_curve(self):
        x, y = self.affine()
        return (-x * x + y * y - 1 - D * x * x * y * y) % L == 0 and self.T * self.Z % L == self.X * self.Y % L

    def normalized(self):
        x, y = self.affine()
        return EdwardsPoint(x, y, 1, x * y)


BASEPOINT = (EdwardsPoint.from_y(8 * pow(9, -1, L) % L) * 8).normalized()
assert BASEPOINT is not None and BASEPOINT.on_curve()
assert BASEPOINT * R_ORDER == EdwardsPoint.identity() and not BASEPOINT == EdwardsPoint.identity()


class RistrettoPoint:
    """zerocaf::ristretto::RistrettoPoint(pub EdwardsPoint)  -- `.0` is `.p` here."""
    __slots__ = ("p",)

    def __init__(self, p):
        self.p = p

    def double(self):
        return RistrettoPoint(self.p.double())

    def __add__(self, o):
        return RistrettoPoint(self.p + o.p)

    def __mul__(self, k):
        return RistrettoPoint(self.p * k)


RISTRETTO_BASEPOINT = RistrettoPoint(BASEPOINT)


def random_point(rnd):
    """stand-in for new_random_point(&mut rng): a random multiple of the basepoint,
    re-randomised projectively (Z != 1) like the outputs of zerocaf arithmetic."""
    p = BASEPOINT * rnd.randrange(1, R_ORDER)
    z = rnd.randrange(1, L)
    return EdwardsPoint(p.X * z, p.Y * z, p.Z * z, p.T * z)


def scalar_bits(k, nbits=256):
    """zerocaf Scalar::into_bits(): little-endian bits, [u8; 256]"""
    return [(k >> i) & 1 for i in range(nbits)]
