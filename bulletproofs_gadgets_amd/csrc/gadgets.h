// gadgets.h -- the reference's gadget library, host side of the product.
// Same names, same argument meaning, same order of constraint-system calls
// (which fixes wire indices and z-powers, hence proof bytes) and the same hard
// errors as /root/reference/src/gadgets/**.  Every function cites its lines.
#pragma once
#include "r1cs.h"

namespace bpg {

// ref: src/gadgets/boolean.rs:12-21 (the vacuous constraint at :16 is kept: it consumes a z power)
inline void binary_constrain_gadget(ConstraintSystem &cs, Variable bit) {
    LC one(Scalar(1));
    LC one_min_bit = one - LC(bit);
    cs.constrain(one_min_bit - one + LC(bit));
    auto m = cs.multiply(one_min_bit, LC(bit));
    cs.constrain(LC(m.o));
}

// ref: src/gadgets/scalar.rs:9-26.  Throws (the reference panics, :13-15) on a zero assignment.
inline void nonzero_gadget(const LC &var, const Scalar *var_assignment, ConstraintSystem &cs) {
    ConstraintSystem::Triple t;
    if (var_assignment) {
        Scalar inverse = var_assignment->invert();
        if (inverse.is_zero()) throw R1CSError(BPG_EGADGET, "Attempting to divide by 0 on an inversion op.");
        Scalar one(1);
        t = cs.allocate_multiplier(&inverse, &one);
    } else {
        t = cs.allocate_multiplier(nullptr, nullptr);
    }
    auto m = cs.multiply(LC(t.l), var);
    cs.constrain(LC(m.o) - LC(Scalar(1)));
}

// ref: src/gadgets/point/edwards_point.rs:10-15
struct SonnyEdwardsPointGadget {
    LC X, Y, Z, T;

    // ref: edwards_point.rs:19-26
    static SonnyEdwardsPointGadget from_point(const SonnyPoint &p) { return {LC(p.X), LC(p.Y), LC(p.Z), LC(p.T)}; }

    // ref: edwards_point.rs:28-93 -- 12 multiply, 6 constrain
    SonnyEdwardsPointGadget add(const SonnyEdwardsPointGadget &other, ConstraintSystem &cs) const {
        const Scalar a = sonny_a(), d = sonny_d();
        auto mA = cs.multiply(X, other.X);            // (X, other_x, A)
        auto mB = cs.multiply(Y, other.Y);            // (Y, other_y, B)
        auto mpt = cs.multiply(T, other.T);
        auto mC = cs.multiply(LC(mpt.o), LC(d));
        auto mD = cs.multiply(Z, other.Z);
        LC E1 = X + LC(mB.l);
        cs.constrain(E1 - LC(mA.l) - LC(mB.l));
        LC E2 = other.X + other.Y;
        cs.constrain(E2 - LC(mA.r) - LC(mB.r));
        auto mE12 = cs.multiply(E1, E2);
        auto maA = cs.multiply(LC(a), LC(mA.o));
        auto mbB = cs.multiply(LC(a), LC(mB.o));
        LC E = LC(maA.o) + LC(mbB.o) + LC(mE12.o);
        cs.constrain(E - LC(maA.o) - LC(mbB.o) - LC(mE12.o));
        LC F = LC(mD.o) - LC(mC.o);
        cs.constrain(F - LC(mD.o) + LC(mC.o));
        LC G = LC(mD.o) + LC(mC.o);
        cs.constrain(G - LC(mD.o) - LC(mC.o));
        LC H = LC(mB.o) + LC(mA.o);
        cs.constrain(H - LC(mB.o) - LC(mA.o));
        auto mX = cs.multiply(E, F);                  // (E, F, X)
        auto mY = cs.multiply(G, H);                  // (G, H, Y)
        auto mZ = cs.multiply(LC(mX.r), LC(mY.l));
        auto mT = cs.multiply(LC(mX.l), LC(mY.r));
        return {LC(mX.o), LC(mY.o), LC(mZ.o), LC(mT.o)};
    }

    // ref: edwards_point.rs:97-137 -- 10 multiply, 4 constrain
    SonnyEdwardsPointGadget double_(ConstraintSystem &cs) const {
        LC a(sonny_a());
        auto mA = cs.multiply(X, X);                  // (X, _, A)
        auto mB = cs.multiply(Y, Y);                  // (Y, _, B)
        auto mzz = cs.multiply(Z, Z);
        auto mC = cs.multiply(LC(Scalar(2)), LC(mzz.o));
        auto mD = cs.multiply(a, LC(mA.o));
        LC xy = LC(mA.l) + LC(mB.l);
        auto msq = cs.multiply(xy, xy);
        LC E = LC(msq.o) - LC(mA.o) - LC(mB.o);
        cs.constrain(E - LC(msq.o) + LC(mA.o) + LC(mB.o));
        LC G = LC(mD.o) + LC(mB.o);
        cs.constrain(G - LC(mD.o) - LC(mB.o));
        LC F = G - LC(mC.o);
        cs.constrain(F - G + LC(mC.o));
        LC H = LC(mD.o) - LC(mB.o);
        cs.constrain(H - LC(mD.o) + LC(mB.o));
        auto oX = cs.multiply(E, F).o;
        auto oY = cs.multiply(G, H).o;
        auto oZ = cs.multiply(F, G).o;
        auto oT = cs.multiply(E, H).o;
        return {LC(oX), LC(oY), LC(oZ), LC(oT)};
    }

    // ref: edwards_point.rs:215-253.  In the Rust, `let (bit, _, ..)` inside the y' and z' blocks shadows `bit`
    // only within each block (:227, :237): the third multiply (bit * Z, :237) and the fourth (T * bit, :246)
    // take the right wire of the FIRST multiply (:221) again; the inner wires only appear in y' and z'.
    SonnyEdwardsPointGadget conditionally_select(const LC &bit_lc, ConstraintSystem &cs) const {
        LC one(Scalar(1));
        auto m1 = cs.multiply(X, bit_lc);                       // (_, bit, x')
        const Variable bit = m1.r;
        auto m2 = cs.multiply(LC(bit), Y);                      // (bit_y, _, bit*y)
        LC y_prime = LC(m2.o) + one - LC(m2.l);
        cs.constrain(y_prime - LC(m2.o) - one + LC(m2.l));
        auto m3 = cs.multiply(LC(bit), Z);                      // (bit_z, _, bit*z)
        LC z_prime = LC(m3.o) + one - LC(m3.l);
        cs.constrain(z_prime - LC(m3.o) - one + LC(m3.l));
        auto m4 = cs.multiply(T, LC(bit));
        return {LC(m1.o), y_prime, z_prime, LC(m4.o)};
    }

    // ref: edwards_point.rs:140-163 -- MSB first (sk.reverse() at :153); 27 multipliers per bit
    static SonnyEdwardsPointGadget scalar_mul(const SonnyEdwardsPointGadget &point, const std::vector<Variable> &sk,
                                              ConstraintSystem &cs) {
        SonnyEdwardsPointGadget Q{LC(Scalar(0)), LC(Scalar(1)), LC(Scalar(1)), LC(Scalar(0))};
        for (size_t k = sk.size(); k-- > 0;) {
            Variable var = sk[k];
            binary_constrain_gadget(cs, var);
            Q = Q.double_(cs);
            SonnyEdwardsPointGadget point_or_id = point.conditionally_select(LC(var), cs);
            Q = Q.add(point_or_id, cs);
        }
        return Q;
    }

    // ref: edwards_point.rs:166-174
    void equal(const SonnyEdwardsPointGadget &other, ConstraintSystem &cs) const {
        auto ma = cs.multiply(X, other.Z);            // (_, other_z, a)
        auto mb = cs.multiply(other.X, Z);            // (_, Z, b)
        cs.constrain(LC(ma.o) - LC(mb.o));
        auto mc = cs.multiply(Y, LC(ma.r));
        auto md = cs.multiply(other.Y, LC(mb.r));
        cs.constrain(LC(mc.o) - LC(md.o));
    }

    // ref: edwards_point.rs:178-211
    void satisfy_curve_eq(ConstraintSystem &cs) const {
        LC a(sonny_a()), d(sonny_d());
        auto xsq = cs.multiply(X, X).o;
        auto axsq = cs.multiply(a, LC(xsq)).o;
        auto ysq = cs.multiply(Y, Y).o;
        LC ax_sq_y_sq = LC(axsq) + LC(ysq);
        cs.constrain(ax_sq_y_sq - LC(axsq) - LC(ysq));
        auto zsq = cs.multiply(Z, Z).o;
        auto left = cs.multiply(ax_sq_y_sq, LC(zsq)).o;
        auto zss = cs.multiply(LC(zsq), LC(zsq)).o;
        auto dxsq = cs.multiply(d, LC(xsq)).o;
        auto dxy = cs.multiply(LC(dxsq), LC(ysq)).o;
        LC right = LC(zss) + LC(dxy);
        cs.constrain(right - LC(zss) - LC(dxy));
        cs.constrain(right - LC(left));
    }
};

// ref: src/gadgets/point/ristretto_point.rs:12-17
struct SonnyRistrettoPointGadget {
    LC X, Y, Z, T;

    // ref: ristretto_point.rs:92-168 -- 12 multiply, 6 constrain; E1/E2 from the raw LCs (:127-131)
    SonnyRistrettoPointGadget add(ConstraintSystem &cs, const SonnyRistrettoPointGadget &other) const {
        const Scalar a = sonny_a(), d = sonny_d();
        auto A = cs.multiply(X, other.X).o;
        auto B = cs.multiply(Y, other.Y).o;
        auto pt = cs.multiply(T, other.T).o;
        auto C = cs.multiply(LC(pt), LC(d)).o;
        auto D = cs.multiply(Z, other.Z).o;
        LC E1 = X + Y;
        cs.constrain(E1 - X - Y);
        LC E2 = other.X + other.Y;
        cs.constrain(E2 - other.X - other.Y);
        auto E12 = cs.multiply(E1, E2).o;
        auto aA = cs.multiply(LC(a), LC(A)).o;
        auto bB = cs.multiply(LC(a), LC(B)).o;
        LC E = LC(aA) + LC(bB) + LC(E12);
        cs.constrain(E - LC(aA) - LC(bB) - LC(E12));
        LC F = LC(D) - LC(C);
        cs.constrain(F - LC(D) + LC(C));
        LC G = LC(D) + LC(C);
        cs.constrain(G - LC(D) - LC(C));
        LC H = LC(B) + LC(A);
        cs.constrain(H - LC(B) - LC(A));
        auto mX = cs.multiply(E, F);
        auto mY = cs.multiply(G, H);
        auto oZ = cs.multiply(LC(mX.r), LC(mY.l)).o;
        auto oT = cs.multiply(LC(mX.l), LC(mY.r)).o;
        return {LC(mX.o), LC(mY.o), LC(oZ), LC(oT)};
    }

    // ref: ristretto_point.rs:177-180 -- add(self, self): 12 multipliers
    SonnyRistrettoPointGadget double_(ConstraintSystem &cs) const { return add(cs, *this); }

    // ref: ristretto_point.rs:171-175
    void equals(ConstraintSystem &cs, const SonnyRistrettoPointGadget &other) const {
        auto x1y2 = cs.multiply(X, other.Y).o;
        auto y1x2 = cs.multiply(Y, other.X).o;
        cs.constrain(LC(x1y2) - LC(y1x2));
    }

    // ref: ristretto_point.rs:184-226 (same body as the Edwards gadget's)
    SonnyRistrettoPointGadget conditionally_select(const LC &bit, ConstraintSystem &cs) const {
        SonnyEdwardsPointGadget e{X, Y, Z, T};
        auto r = e.conditionally_select(bit, cs);
        return {r.X, r.Y, r.Z, r.T};
    }

    // ref: ristretto_point.rs:50-90 (incl. the sign difference between :72 and :86)
    void ristretto_gadget(ConstraintSystem &cs, const SonnyPoint *point_assign) const {
        SonnyRistrettoPointGadget eight_p = double_(cs).double_(cs).double_(cs);
        if (point_assign) {
            SonnyPoint p8 = sonny_double(sonny_double(sonny_double(*point_assign)));
            nonzero_gadget(eight_p.X, &p8.X, cs);
            LC y_m_z = eight_p.Y - eight_p.Z;
            cs.constrain(eight_p.Y - eight_p.Z - y_m_z);
            Scalar ymz = p8.Y - p8.Z;
            nonzero_gadget(y_m_z, &ymz, cs);
        } else {
            nonzero_gadget(eight_p.X, nullptr, cs);
            LC y_m_z = eight_p.Y - eight_p.Z;
            cs.constrain(eight_p.Y - eight_p.Z + y_m_z);
            nonzero_gadget(y_m_z, nullptr, cs);
        }
    }

    // ref: ristretto_point.rs:23-32
    static SonnyRistrettoPointGadget from_point(const SonnyPoint &p, ConstraintSystem &cs) {
        SonnyRistrettoPointGadget g{LC(p.X), LC(p.Y), LC(p.Z), LC(p.T)};
        g.ristretto_gadget(cs, &p);
        return g;
    }

    // ref: ristretto_point.rs:34-45 (asserts 4 LCs)
    static SonnyRistrettoPointGadget from_lcs(const std::vector<LC> &lcs, ConstraintSystem &cs) {
        if (lcs.size() != 4) throw R1CSError(BPG_EGADGET, "from_lcs: expected 4 linear combinations");
        SonnyRistrettoPointGadget g{lcs[0], lcs[1], lcs[2], lcs[3]};
        g.ristretto_gadget(cs, nullptr);
        return g;
    }
};

// ref: src/gadgets/sk_knowledge/sk_know_gadget.rs:13-38
inline void sk_knowledge_gadget(ConstraintSystem &cs, const SonnyRistrettoPointGadget &basep,
                                const SonnyRistrettoPointGadget &pk, const std::vector<Variable> &sk) {
    SonnyRistrettoPointGadget Q{LC(Scalar(0)), LC(Scalar(1)), LC(Scalar(1)), LC(Scalar(0))};
    for (size_t k = sk.size(); k-- > 0;) {
        Variable var = sk[k];
        binary_constrain_gadget(cs, var);
        Q = Q.double_(cs);
        SonnyRistrettoPointGadget basep_or_id = basep.conditionally_select(LC(var), cs);
        Q = Q.add(cs, basep_or_id);
    }
    pk.equals(cs, Q);
}

}  // namespace bpg
