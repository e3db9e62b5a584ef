"""oracle/gadgets.py -- TEST INFRASTRUCTURE.

Restatement, quirk for quirk, of the reference's gadget library
(/root/reference/src/gadgets/**), written against the oracle's constraint-system
interface (oracle/r1cs.py: multiply / allocate_multiplier / constrain).  Each
function cites the reference lines it follows.  The order of cs calls is what
fixes the circuit (wire indices and z-powers), so it is kept exactly.
"""
from . import sonny
from .r1cs import LC, L, inv

A = sonny.A            # zerocaf::constants::EDWARDS_A as a Scalar
D = sonny.D            # zerocaf::constants::EDWARDS_D as a Scalar


def binary_constrain_gadget(cs, bit):
    """ref: src/gadgets/boolean.rs:12-21 (incl. the vacuous constraint at :16)"""
    one = LC.of(1)
    one_min_bit = one.clone() - bit
    cs.constrain(one_min_bit.clone() - one + bit)
    _, _, res = cs.multiply(one_min_bit, LC.of(bit))
    cs.constrain(LC.of(res))


def nonzero_gadget(var, var_assignment, cs):
    """ref: src/gadgets/scalar.rs:9-26"""
    assignment = None
    if var_assignment is not None:
        inverse = inv(var_assignment)
        if inverse == 0:
            raise RuntimeError("Attempting to divide by 0 on an inversion op.")    # panic!, scalar.rs:13-15
        assignment = (inverse, 1)
    inv_var, _, _ = cs.allocate_multiplier(assignment)
    _, _, should_be_one = cs.multiply(LC.of(inv_var), var)
    cs.constrain(should_be_one - LC.of(1))


class SonnyEdwardsPointGadget:
    """ref: src/gadgets/point/edwards_point.rs:10-15"""

    def __init__(self, X, Y, Z, T):
        self.X, self.Y, self.Z, self.T = LC.of(X), LC.of(Y), LC.of(Z), LC.of(T)

    @staticmethod
    def from_point(p):
        """ref: edwards_point.rs:19-26"""
        return SonnyEdwardsPointGadget(p.X, p.Y, p.Z, p.T)

    def add(self, other, cs):
        """ref: edwards_point.rs:28-93 -- 12 multiply, 6 constrain"""
        X, other_x, A_ = cs.multiply(self.X.clone(), other.X.clone())
        Y, other_y, B_ = cs.multiply(self.Y.clone(), other.Y.clone())
        _, _, pt = cs.multiply(self.T.clone(), other.T.clone())
        _, _, C = cs.multiply(LC.of(pt), LC.of(D))
        _, _, D_ = cs.multiply(self.Z.clone(), other.Z.clone())
        E1 = self.X.clone() + Y
        cs.constrain(E1.clone() - X - Y)
        E2 = other.X.clone() + other.Y.clone()
        cs.constrain(E2.clone() - other_x - other_y)
        _, _, E12 = cs.multiply(E1, E2)
        _, _, aA = cs.multiply(LC.of(A), LC.of(A_))
        _, _, bB = cs.multiply(LC.of(A), LC.of(B_))
        E = aA + bB + E12
        cs.constrain(E.clone() - aA - bB - E12)
        F = D_ - C
        cs.constrain(F.clone() - D_ + C)
        G = D_ + C
        cs.constrain(G.clone() - D_ - C)
        H = B_ + A_
        cs.constrain(H.clone() - B_ - A_)
        E, F, X = cs.multiply(E, F)
        G, H, Y = cs.multiply(G, H)
        _, _, Z = cs.multiply(LC.of(F), LC.of(G))
        _, _, T = cs.multiply(LC.of(E), LC.of(H))
        return SonnyEdwardsPointGadget(X, Y, Z, T)

    def double(self, cs):
        """ref: edwards_point.rs:97-137 -- 10 multiply, 4 constrain"""
        a = LC.of(A)
        X, _, A_ = cs.multiply(self.X.clone(), self.X.clone())
        Y, _, B_ = cs.multiply(self.Y.clone(), self.Y.clone())
        z_sq = cs.multiply(self.Z.clone(), self.Z.clone())[2]
        C = cs.multiply(LC.of(2), LC.of(z_sq))[2]
        D_ = cs.multiply(a, LC.of(A_))[2]
        p1xy_sq = cs.multiply(X + Y, X + Y)[2]
        E = p1xy_sq - A_ - B_
        cs.constrain(E.clone() - p1xy_sq + A_ + B_)
        G = D_ + B_
        cs.constrain(G.clone() - D_ - B_)
        F = G.clone() - C
        cs.constrain(F.clone() - G.clone() + C)
        H = D_ - B_
        cs.constrain(H.clone() - D_ + B_)
        return SonnyEdwardsPointGadget(
            LC.of(cs.multiply(E.clone(), F.clone())[2]),
            LC.of(cs.multiply(G.clone(), H.clone())[2]),
            LC.of(cs.multiply(F, G)[2]),
            LC.of(cs.multiply(E, H)[2]))

    @staticmethod
    def scalar_mul(point, sk, cs):
        """ref: edwards_point.rs:140-163 -- MSB first (sk.reverse() at :153), 27 multipliers / bit"""
        Q = SonnyEdwardsPointGadget(0, 1, 1, 0)
        for var in reversed(list(sk)):
            binary_constrain_gadget(cs, var)
            Q = Q.double(cs)
            point_or_id = point.conditionally_select(LC.of(var), cs)
            Q = Q.add(point_or_id, cs)
        return Q

    def equal(self, other, cs):
        """ref: edwards_point.rs:166-174"""
        _, other_z, a = cs.multiply(self.X.clone(), other.Z.clone())
        _, Z, b = cs.multiply(other.X.clone(), self.Z.clone())
        cs.constrain(a - b)
        _, _, c = cs.multiply(self.Y.clone(), LC.of(other_z))
        _, _, d = cs.multiply(other.Y.clone(), LC.of(Z))
        cs.constrain(c - d)

    def satisfy_curve_eq(self, cs):
        """ref: edwards_point.rs:178-211"""
        a, d = LC.of(A), LC.of(D)
        _, _, x_sq = cs.multiply(self.X.clone(), self.X.clone())
        _, _, aX_sq = cs.multiply(a, LC.of(x_sq))
        _, _, y_sq = cs.multiply(self.Y.clone(), self.Y.clone())
        ax_sq_y_sq = aX_sq + y_sq
        cs.constrain(ax_sq_y_sq.clone() - aX_sq - y_sq)
        _, _, z_sq = cs.multiply(self.Z.clone(), self.Z.clone())
        _, _, left = cs.multiply(ax_sq_y_sq, LC.of(z_sq))
        _, _, z_s_s = cs.multiply(LC.of(z_sq), LC.of(z_sq))
        _, _, dx_sq = cs.multiply(d, LC.of(x_sq))
        _, _, dx_sq_y_sq = cs.multiply(LC.of(dx_sq), LC.of(y_sq))
        right = z_s_s + dx_sq_y_sq
        cs.constrain(right.clone() - z_s_s - dx_sq_y_sq)
        cs.constrain(right - left)

    def conditionally_select(self, bit, cs):
        """ref: edwards_point.rs:215-253.  The `let (bit, _, ..)` bindings at :227 and :237 live inside the
        y' / z' blocks only: the multiplies at :237 and :246 use the right wire of the first multiply (:221)."""
        one = LC.of(1)
        _, bit, x_prime = cs.multiply(self.X.clone(), bit)
        bit_y, _, bit_t_y = cs.multiply(LC.of(bit), self.Y.clone())
        y_prime = bit_t_y + one.clone() - bit_y
        cs.constrain(y_prime.clone() - bit_t_y - one.clone() + bit_y)
        bit_z, _, bit_t_z = cs.multiply(LC.of(bit), self.Z.clone())
        z_prime = bit_t_z + one.clone() - bit_z
        cs.constrain(z_prime.clone() - bit_t_z - one + bit_z)
        _, _, t_prime = cs.multiply(self.T.clone(), LC.of(bit))
        return type(self)(x_prime, y_prime, z_prime, t_prime)

    @staticmethod
    def prover_commit_to_sonny_edwards_point(prover, p, rng):
        """ref: edwards_point.rs:255-276 (Scalar::random(&mut thread_rng()) -> the injected rng)"""
        commitments, vars_ = [], []
        for x in (p.X, p.Y, p.Z, p.T):
            V, var = prover.commit(x, rng.scalar())
            commitments.append(V)
            vars_.append(var)
        return SonnyEdwardsPointGadget(*vars_), commitments

    @staticmethod
    def verifier_commit_to_sonny_edwards_point(verifier, commitments):
        """ref: edwards_point.rs:278-290"""
        assert len(commitments) == 4
        vars_ = [verifier.commit(V) for V in commitments]
        return SonnyEdwardsPointGadget(*vars_)


class SonnyRistrettoPointGadget:
    """ref: src/gadgets/point/ristretto_point.rs:12-17"""

    def __init__(self, X, Y, Z, T):
        self.X, self.Y, self.Z, self.T = LC.of(X), LC.of(Y), LC.of(Z), LC.of(T)

    def clone(self):
        return SonnyRistrettoPointGadget(self.X.clone(), self.Y.clone(), self.Z.clone(), self.T.clone())

    @staticmethod
    def from_point(point, cs):
        """ref: ristretto_point.rs:23-32"""
        g = SonnyRistrettoPointGadget(point.p.X, point.p.Y, point.p.Z, point.p.T)
        g.ristretto_gadget(cs, point)
        return g

    @staticmethod
    def from_lcs(lcs, cs):
        """ref: ristretto_point.rs:34-45"""
        assert len(lcs) == 4
        g = SonnyRistrettoPointGadget(*[LC.of(x).clone() for x in lcs])
        g.ristretto_gadget(cs, None)
        return g

    def ristretto_gadget(self, cs, point_assign):
        """ref: ristretto_point.rs:50-90 (incl. the sign difference between :72 and :86)"""
        eight_p = self.double(cs).double(cs).double(cs)
        if point_assign is not None:
            p8 = point_assign.double().double().double()
            nonzero_gadget(eight_p.X, p8.p.X, cs)
            y_m_z = eight_p.Y.clone() - eight_p.Z.clone()
            cs.constrain(eight_p.Y.clone() - eight_p.Z - y_m_z.clone())
            nonzero_gadget(y_m_z, (p8.p.Y - p8.p.Z) % L, cs)
        else:
            nonzero_gadget(eight_p.X, None, cs)
            y_m_z = eight_p.Y.clone() - eight_p.Z.clone()
            cs.constrain(eight_p.Y.clone() - eight_p.Z + y_m_z.clone())
            nonzero_gadget(y_m_z, None, cs)

    def add(self, cs, other):
        """ref: ristretto_point.rs:92-168 -- 12 multiply, 6 constrain; E1/E2 from the raw LCs (:127-131)"""
        _, _, A_ = cs.multiply(self.X.clone(), other.X.clone())
        _, _, B_ = cs.multiply(self.Y.clone(), other.Y.clone())
        _, _, pt = cs.multiply(self.T, other.T)
        _, _, C = cs.multiply(LC.of(pt), LC.of(D))
        _, _, D_ = cs.multiply(self.Z, other.Z)
        E1 = self.X.clone() + self.Y.clone()
        cs.constrain(E1.clone() - self.X - self.Y)
        E2 = other.X.clone() + other.Y.clone()
        cs.constrain(E2.clone() - other.X - other.Y)
        _, _, E12 = cs.multiply(E1, E2)
        _, _, aA = cs.multiply(LC.of(A), LC.of(A_))
        _, _, bB = cs.multiply(LC.of(A), LC.of(B_))
        E = aA + bB + E12
        cs.constrain(E.clone() - aA - bB - E12)
        F = D_ - C
        cs.constrain(F.clone() - D_ + C)
        G = D_ + C
        cs.constrain(G.clone() - D_ - C)
        H = B_ + A_
        cs.constrain(H.clone() - B_ - A_)
        E, F, X = cs.multiply(E, F)
        G, H, Y = cs.multiply(G, H)
        _, _, Z = cs.multiply(LC.of(F), LC.of(G))
        _, _, T = cs.multiply(LC.of(E), LC.of(H))
        return SonnyRistrettoPointGadget(X, Y, Z, T)

    def equals(self, cs, other):
        """ref: ristretto_point.rs:171-175"""
        _, _, x1y2 = cs.multiply(self.X.clone(), other.Y)
        _, _, y1x2 = cs.multiply(self.Y.clone(), other.X)
        cs.constrain(x1y2 - y1x2)

    def double(self, cs):
        """ref: ristretto_point.rs:177-180 -- add(self, self): 12 multipliers"""
        return self.clone().add(cs, self.clone())

    conditionally_select = SonnyEdwardsPointGadget.conditionally_select     # ref: ristretto_point.rs:184-226, identical body


def sk_knowledge_gadget(cs, basep, pk, sk):
    """ref: src/gadgets/sk_knowledge/sk_know_gadget.rs:13-38"""
    Q = SonnyRistrettoPointGadget(0, 1, 1, 0)
    for var in reversed(list(sk)):
        binary_constrain_gadget(cs, var)
        Q = Q.double(cs)
        basep_or_id = basep.conditionally_select(LC.of(var), cs)
        Q = Q.add(cs, basep_or_id)
    pk.equals(cs, Q)
