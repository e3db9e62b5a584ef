"""oracle/r1cs.py -- TEST INFRASTRUCTURE (see oracle/oracle.h).

CPU restatement of the `bulletproofs::r1cs` protocol that the reference calls
(git dependency dusk-network/bulletproofs@develop with feature `yoloproofs`,
ref: Cargo.toml:15-18; no lockfile, sources absent -- restated from the published
bulletproofs 2.0 algorithm, UPSTREAM-RECALL, see DESIGN.md "parity pins"):

    Prover::new / commit / multiply / allocate_multiplier / constrain / prove
        ref call sites: tests/edwards_point.rs:28,31,47;  src/gadgets/scalar.rs:10-23,46
    Verifier::new / commit / ... / verify
        ref call sites: tests/edwards_point.rs:61,63,79
    InnerProductProof::create / verification_scalars   (inside prove / verify)
    R1CSProof::to_bytes layout (1 version byte | 11 or 14 x 32 B | L0 R0 ... | a | b)

Scalars are Python ints mod l; every group operation, the Merlin transcript and
the TranscriptRng run in oracle/liboracle.so with dalek's schedules (constant-time
Straus for A_I/A_O/S and Pedersen commitments, vartime Straus/Pippenger for the
IPP and the verifier, one 2-term vartime MSM per folded generator).
"""
import ctypes

from . import binding

L = binding.L


def sc(x):
    return (x % L).to_bytes(32, "little")


def sc_join(xs):
    return b"".join((x % L).to_bytes(32, "little") for x in xs)


def inv(x):
    return pow(x % L, L - 2, L)      # Scalar::invert: 0 -> 0, like dalek


# ----------------------------------------------------------------------------- variables / LCs

class Variable:
    """bulletproofs::r1cs::Variable: Committed(i) | MultiplierLeft(i) | MultiplierRight(i) | MultiplierOutput(i) | One()"""
    __slots__ = ("kind", "i")

    def __init__(self, kind, i=0):
        self.kind, self.i = kind, i

    def _lc(self):
        return LinearCombination([(self, 1)])

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

    def __repr__(self):
        return "%s(%d)" % (self.kind, self.i)


ONE = Variable("One")


class LinearCombination:
    """Vec<(Variable, Scalar)> with From<Scalar | Variable> and Add / Sub / Neg."""
    __slots__ = ("terms",)

    def __init__(self, terms=None):
        self.terms = list(terms or [])

    @staticmethod
    def of(x):
        if isinstance(x, LinearCombination):
            return x
        if isinstance(x, Variable):
            return LinearCombination([(x, 1)])
        return LinearCombination([(ONE, int(x) % L)])

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


LC = LinearCombination


class R1CSError(Exception):
    pass


class VerificationError(R1CSError):
    pass


class InvalidGeneratorsLength(R1CSError):
    pass


class MissingAssignment(R1CSError):
    pass


# ----------------------------------------------------------------------------- generators / transcript

class PedersenGens:
    def __init__(self, orc):
        self.orc = orc
        self.B, self.B_blinding = orc.pedersen()
        self.both, _ = orc.concat((self.B, 1), (self.B_blinding, 1))

    def commit(self, v, r):
        """compress(v*B + r*B_blinding) (RistrettoPoint::multiscalar_mul)"""
        return self.orc.msm(sc(v) + sc(r), self.both, 2, binding.SCHED_STRAUS_CT)


class BulletproofGens:
    def __init__(self, orc, gens_capacity, party_capacity=1):
        assert party_capacity == 1
        self.orc, self.gens_capacity = orc, gens_capacity
        self.G, self.H = orc.gens(gens_capacity)


class Transcript:
    """merlin::Transcript + bulletproofs' TranscriptProtocol extension."""

    def __init__(self, orc, label):
        self.orc = orc
        self.st = ctypes.create_string_buffer(256)
        orc.lib.merlin_init(self.st, label, len(label))

    def append_message(self, label, msg):
        self.orc.lib.merlin_append_message(self.st, label, msg, len(msg))

    def append_u64(self, label, x):
        self.append_message(label, int(x).to_bytes(8, "little"))

    def append_scalar(self, label, x):
        self.append_message(label, sc(x))

    def append_point(self, label, p32):
        self.append_message(label, p32)

    def validate_and_append_point(self, label, p32):
        if p32 == bytes(32):
            raise VerificationError("identity point")
        self.append_message(label, p32)

    def challenge_scalar(self, label):
        out = ctypes.create_string_buffer(32)
        self.orc.lib.merlin_challenge_scalar(self.st, label, out)
        return int.from_bytes(out.raw, "little")

    def r1cs_domain_sep(self):
        self.append_message(b"dom-sep", b"r1cs v1")

    def r1cs_1phase_domain_sep(self):
        self.append_message(b"dom-sep", b"r1cs-1phase")

    def r1cs_2phase_domain_sep(self):
        self.append_message(b"dom-sep", b"r1cs-2phase")

    def innerproduct_domain_sep(self, n):
        self.append_message(b"dom-sep", b"ipp v1")
        self.append_u64(b"n", n)

    def build_rng(self, witness_scalars, external32):
        """build_rng().rekey_with_witness_bytes("v_blinding", ..)*.finalize(external rng)"""
        r = ctypes.create_string_buffer(256)
        self.orc.lib.merlin_rng_build(r, self.st)
        for w in witness_scalars:
            self.orc.lib.merlin_rng_rekey(r, b"v_blinding", sc(w), 32)
        self.orc.lib.merlin_rng_finalize(r, external32)
        return TranscriptRng(self.orc, r)


class TranscriptRng:
    def __init__(self, orc, st):
        self.orc, self.st = orc, st

    def scalar(self):
        out = ctypes.create_string_buffer(32)
        self.orc.lib.merlin_rng_scalar(self.st, out)
        return int.from_bytes(out.raw, "little")

    def scalars(self, n):
        out = ctypes.create_string_buffer(32 * max(n, 1))
        self.orc.lib.merlin_rng_scalars(self.st, out, ctypes.c_size_t(n))
        raw = out.raw                      # one copy (out.raw copies the whole buffer every time it is read)
        return [int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(n)]


class SeededRng:
    """Deterministic stand-in for rand::thread_rng(): a SHAKE256 stream (seed | "rng")."""

    def __init__(self, seed=b"bpg-bench-v1"):
        import hashlib
        self._x = hashlib.shake_256(seed + b"rng")
        self._pos = 0

    def fill_bytes(self, n):
        out = self._x.digest(self._pos + n)[self._pos:]
        self._pos += n
        return out

    def scalar(self):
        """Scalar::random(&mut rng)"""
        return int.from_bytes(self.fill_bytes(64), "little") % L


# ----------------------------------------------------------------------------- proof object

class InnerProductProof:
    def __init__(self, L_vec, R_vec, a, b):
        self.L_vec, self.R_vec, self.a, self.b = L_vec, R_vec, a, b


class R1CSProof:
    FIELDS = ("A_I1", "A_O1", "S1", "A_I2", "A_O2", "S2", "T_1", "T_3", "T_4", "T_5", "T_6")

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def to_bytes(self):
        one_phase = self.A_I2 == bytes(32) and self.A_O2 == bytes(32) and self.S2 == bytes(32)
        out = bytes([0 if one_phase else 1]) + self.A_I1 + self.A_O1 + self.S1
        if not one_phase:
            out += self.A_I2 + self.A_O2 + self.S2
        out += self.T_1 + self.T_3 + self.T_4 + self.T_5 + self.T_6
        out += sc(self.t_x) + sc(self.t_x_blinding) + sc(self.e_blinding)
        for l, r in zip(self.ipp.L_vec, self.ipp.R_vec):
            out += l + r
        return out + sc(self.ipp.a) + sc(self.ipp.b)

    @staticmethod
    def from_bytes(b):
        if len(b) < 1 or b[0] not in (0, 1):
            raise R1CSError("format")
        k = 11 if b[0] == 0 else 14
        body = b[1:]
        if len(body) % 32 or len(body) < (k + 2) * 32 or ((len(body) // 32 - k - 2) % 2):
            raise R1CSError("format")
        ch = [body[32 * i:32 * i + 32] for i in range(len(body) // 32)]
        f = {}
        names = ["A_I1", "A_O1", "S1"] + (["A_I2", "A_O2", "S2"] if k == 14 else []) + ["T_1", "T_3", "T_4", "T_5", "T_6"]
        for n_, v in zip(names, ch):
            f[n_] = v
        if k == 11:
            f["A_I2"] = f["A_O2"] = f["S2"] = bytes(32)
        i = len(names)
        ints = [int.from_bytes(x, "little") for x in ch[i:i + 3]]
        if any(x >= L for x in ints):
            raise R1CSError("format")
        f["t_x"], f["t_x_blinding"], f["e_blinding"] = ints
        rest = ch[i + 3:]
        a, bb = int.from_bytes(rest[-2], "little"), int.from_bytes(rest[-1], "little")
        if a >= L or bb >= L:
            raise R1CSError("format")
        f["ipp"] = InnerProductProof(rest[0:-2:2], rest[1:-2:2], a, bb)
        return R1CSProof(**f)


# ----------------------------------------------------------------------------- constraint systems

class _CS:
    """ConstraintSystem + RandomizableConstraintSystem + RandomizedConstraintSystem (bulletproofs::r1cs, feature
    yoloproofs).  `RandomizedConstraintSystem` is imported by the reference at src/gadgets/point/ristretto_point.rs:3;
    none of its gadgets registers a randomized constraint, so this half of the protocol is exercised by a
    shuffle-style circuit of the tests only."""
    _phase2 = False

    def specify_randomized_constraints(self, callback):
        """callback(cs) runs after the first-phase commitments A_I1, A_O1, S1 are in the transcript; it may draw
        challenges (cs.challenge_scalar), allocate multipliers and add constraints"""
        self.deferred.append(callback)

    def challenge_scalar(self, label):
        """RandomizedConstraintSystem::challenge_scalar: only inside a randomized-constraints callback"""
        if not self._phase2:
            raise R1CSError("challenge_scalar outside the randomization phase")
        return self.transcript.challenge_scalar(label)

    def _create_randomized_constraints(self):
        if not self.deferred:
            self.transcript.r1cs_1phase_domain_sep()
            return
        self.transcript.r1cs_2phase_domain_sep()
        callbacks, self.deferred = self.deferred, []
        self._phase2 = True
        try:
            for cb in callbacks:
                cb(self)
        finally:
            self._phase2 = False

    def multiply(self, left, right):
        raise NotImplementedError

    def allocate_multiplier(self, assignment):
        raise NotImplementedError

    def constrain(self, lc):
        self.constraints.append(LinearCombination.of(lc))

    def _flatten(self, z, n, m, want_wc):
        wL, wR, wO, wV, wc = [0] * n, [0] * n, [0] * n, [0] * m, 0
        exp_z = z
        for lc in self.constraints:
            for var, coeff in lc.terms:
                k = var.kind
                if k == "L":
                    wL[var.i] = (wL[var.i] + exp_z * coeff) % L
                elif k == "R":
                    wR[var.i] = (wR[var.i] + exp_z * coeff) % L
                elif k == "O":
                    wO[var.i] = (wO[var.i] + exp_z * coeff) % L
                elif k == "C":
                    wV[var.i] = (wV[var.i] - exp_z * coeff) % L
                elif want_wc:
                    wc = (wc - exp_z * coeff) % L
            exp_z = exp_z * z % L
        return wL, wR, wO, wV, wc


class Prover(_CS):
    def __init__(self, orc, pc_gens, transcript):
        self.orc, self.pc_gens, self.transcript = orc, pc_gens, transcript
        transcript.r1cs_domain_sep()
        self.constraints, self.a_L, self.a_R, self.a_O, self.v, self.v_blinding = [], [], [], [], [], []
        self.deferred = []

    def commit(self, v, v_blinding):
        i = len(self.v)
        self.v.append(v % L)
        self.v_blinding.append(v_blinding % L)
        V = self.pc_gens.commit(v, v_blinding)
        self.transcript.append_point(b"V", V)
        return V, Variable("C", i)

    def eval(self, lc):
        s = 0
        for var, c in lc.terms:
            k = var.kind
            val = (self.a_L[var.i] if k == "L" else self.a_R[var.i] if k == "R" else self.a_O[var.i] if k == "O"
                   else self.v[var.i] if k == "C" else 1)
            s += c * val
        return s % L

    def multiply(self, left, right):
        left, right = LinearCombination.of(left), LinearCombination.of(right)
        l, r = self.eval(left), self.eval(right)
        i = len(self.a_L)
        lv, rv, ov = Variable("L", i), Variable("R", i), Variable("O", i)
        self.a_L.append(l)
        self.a_R.append(r)
        self.a_O.append(l * r % L)
        self.constrain(left - lv)
        self.constrain(right - rv)
        return lv, rv, ov

    def allocate_multiplier(self, assignment):
        if assignment is None:
            raise MissingAssignment()
        l, r = assignment
        i = len(self.a_L)
        self.a_L.append(l % L)
        self.a_R.append(r % L)
        self.a_O.append(l * r % L)
        return Variable("L", i), Variable("R", i), Variable("O", i)

    def prove(self, bp_gens, external_rng32=bytes(32)):
        orc, tr = self.orc, self.transcript
        tr.append_u64(b"m", len(self.v))
        rng = tr.build_rng(self.v_blinding, external_rng32)
        n1 = len(self.a_L)
        if bp_gens.gens_capacity < n1:
            raise InvalidGeneratorsLength()
        i_bl1, o_bl1, s_bl1 = rng.scalar(), rng.scalar(), rng.scalar()
        s_L1 = rng.scalars(n1)
        s_R1 = rng.scalars(n1)
        CT = binding.SCHED_STRAUS_CT

        def commit3(lo, hi, bl, s_L, s_R):
            """A_I, A_O, S over the multipliers [lo, hi) and the generators G[lo..hi), H[lo..hi)"""
            k = hi - lo
            Gk, Hk = orc.slice(bp_gens.G, lo, k), orc.slice(bp_gens.H, lo, k)
            bgh, _ = orc.concat((self.pc_gens.B_blinding, 1), (Gk, k), (Hk, k))
            bg, _ = orc.concat((self.pc_gens.B_blinding, 1), (Gk, k))
            A_I = orc.msm(sc(bl[0]) + sc_join(self.a_L[lo:hi]) + sc_join(self.a_R[lo:hi]), bgh, 2 * k + 1, CT)
            A_O = orc.msm(sc(bl[1]) + sc_join(self.a_O[lo:hi]), bg, k + 1, CT)
            S = orc.msm(sc(bl[2]) + sc_join(s_L) + sc_join(s_R), bgh, 2 * k + 1, CT)
            return A_I, A_O, S
        A_I1, A_O1, S1 = commit3(0, n1, (i_bl1, o_bl1, s_bl1), s_L1, s_R1)
        tr.append_point(b"A_I1", A_I1)
        tr.append_point(b"A_O1", A_O1)
        tr.append_point(b"S1", S1)
        self._create_randomized_constraints()     # dom-sep r1cs-1phase, or r1cs-2phase + the deferred callbacks
        n = len(self.a_L)
        n2 = n - n1
        padded_n = 1 if n == 0 else 1 << (n - 1).bit_length()
        pad = padded_n - n
        if bp_gens.gens_capacity < padded_n:
            raise InvalidGeneratorsLength()
        ident = bytes(32)
        if n2 > 0:
            i_bl2, o_bl2, s_bl2 = rng.scalar(), rng.scalar(), rng.scalar()
            s_L2 = rng.scalars(n2)
            s_R2 = rng.scalars(n2)
            A_I2, A_O2, S2 = commit3(n1, n, (i_bl2, o_bl2, s_bl2), s_L2, s_R2)
        else:
            i_bl2 = o_bl2 = s_bl2 = 0
            s_L2, s_R2 = [], []
            A_I2 = A_O2 = S2 = ident
        s_L, s_R = s_L1 + s_L2, s_R1 + s_R2
        tr.append_point(b"A_I2", A_I2)
        tr.append_point(b"A_O2", A_O2)
        tr.append_point(b"S2", S2)
        y = tr.challenge_scalar(b"y")
        z = tr.challenge_scalar(b"z")
        wL, wR, wO, wV, _ = self._flatten(z, n, len(self.v), False)
        y_inv = inv(y)
        exp_y_inv = [1] * padded_n
        for i in range(1, padded_n):
            exp_y_inv[i] = exp_y_inv[i - 1] * y_inv % L
        l1, l2, l3 = [0] * n, [0] * n, [0] * n
        r0, r1, r3 = [0] * n, [0] * n, [0] * n
        exp_y = 1
        for i in range(n):
            l1[i] = (self.a_L[i] + exp_y_inv[i] * wR[i]) % L
            l2[i] = self.a_O[i]
            l3[i] = s_L[i]
            r0[i] = (wO[i] - exp_y) % L
            r1[i] = (exp_y * self.a_R[i] + wL[i]) % L
            r3[i] = exp_y * s_R[i] % L
            exp_y = exp_y * y % L

        def ip(a, b):
            return sum(x * y_ for x, y_ in zip(a, b)) % L
        t1 = ip(l1, r0)
        t2 = (ip(l1, r1) + ip(l2, r0)) % L
        t3 = (ip(l2, r1) + ip(l3, r0)) % L
        t4 = (ip(l1, r3) + ip(l3, r1)) % L
        t5 = ip(l2, r3)
        t6 = ip(l3, r3)
        tb1, tb3, tb4, tb5, tb6 = (rng.scalar() for _ in range(5))
        T_1, T_3, T_4 = self.pc_gens.commit(t1, tb1), self.pc_gens.commit(t3, tb3), self.pc_gens.commit(t4, tb4)
        T_5, T_6 = self.pc_gens.commit(t5, tb5), self.pc_gens.commit(t6, tb6)
        for lab, T in ((b"T_1", T_1), (b"T_3", T_3), (b"T_4", T_4), (b"T_5", T_5), (b"T_6", T_6)):
            tr.append_point(lab, T)
        u = tr.challenge_scalar(b"u")
        x = tr.challenge_scalar(b"x")
        tb2 = sum(c * vb for c, vb in zip(wV, self.v_blinding)) % L
        t_x = x * (t1 + x * (t2 + x * (t3 + x * (t4 + x * (t5 + x * t6))))) % L
        t_x_blinding = x * (tb1 + x * (tb2 + x * (tb3 + x * (tb4 + x * (tb5 + x * tb6))))) % L
        l_vec = [(x * (l1[i] + x * (l2[i] + x * l3[i]))) % L for i in range(n)] + [0] * pad
        r_vec = [(r0[i] + x * (r1[i] + x * (x * r3[i]))) % L for i in range(n)] + [0] * pad
        for i in range(n, padded_n):
            r_vec[i] = (-exp_y) % L
            exp_y = exp_y * y % L
        i_bl, o_bl, s_bl = (i_bl1 + u * i_bl2) % L, (o_bl1 + u * o_bl2) % L, (s_bl1 + u * s_bl2) % L
        e_blinding = x * (i_bl + x * (o_bl + x * s_bl)) % L
        tr.append_scalar(b"t_x", t_x)
        tr.append_scalar(b"t_x_blinding", t_x_blinding)
        tr.append_scalar(b"e_blinding", e_blinding)
        w = tr.challenge_scalar(b"w")
        Q = orc.ge_array(1)
        wq = ctypes.create_string_buffer(sc(w))
        orc.lib.msm_straus_ct(Q, wq, self.pc_gens.B, ctypes.c_size_t(1))          # Q = w * B
        G_factors = [1] * n1 + [u] * (n2 + pad)
        H_factors = [exp_y_inv[i] * G_factors[i] % L for i in range(padded_n)]
        ipp = ipp_create(orc, tr, Q, G_factors, H_factors, orc.slice(bp_gens.G, 0, padded_n),
                         orc.slice(bp_gens.H, 0, padded_n), l_vec, r_vec)
        return R1CSProof(A_I1=A_I1, A_O1=A_O1, S1=S1, A_I2=A_I2, A_O2=A_O2, S2=S2, T_1=T_1, T_3=T_3, T_4=T_4,
                         T_5=T_5, T_6=T_6, t_x=t_x, t_x_blinding=t_x_blinding, e_blinding=e_blinding, ipp=ipp)


def ipp_create(orc, tr, Q, G_factors, H_factors, G, H, a, b):
    """InnerProductProof::create, with the real per-round generator folds."""
    n = len(a)
    tr.innerproduct_domain_sep(n)
    L_vec, R_vec = [], []
    first = True
    VT = binding.SCHED_VARTIME
    while n != 1:
        n //= 2
        aL, aR, bL, bR = a[:n], a[n:], b[:n], b[n:]
        cL = sum(x * y for x, y in zip(aL, bR)) % L
        cR = sum(x * y for x, y in zip(aR, bL)) % L
        GL, GR, HL, HR = orc.slice(G, 0, n), orc.slice(G, n, n), orc.slice(H, 0, n), orc.slice(H, n, n)
        if first:
            sL = [aL[i] * G_factors[n + i] % L for i in range(n)] + [bR[i] * H_factors[i] % L for i in range(n)] + [cL]
            sR = [aR[i] * G_factors[i] % L for i in range(n)] + [bL[i] * H_factors[n + i] % L for i in range(n)] + [cR]
        else:
            sL, sR = aL + bR + [cL], aR + bL + [cR]
        ptsL, cnt = orc.concat((GR, n), (HL, n), (Q, 1))
        ptsR, _ = orc.concat((GL, n), (HR, n), (Q, 1))
        Lp = orc.msm(sc_join(sL), ptsL, cnt, VT)
        Rp = orc.msm(sc_join(sR), ptsR, cnt, VT)
        L_vec.append(Lp)
        R_vec.append(Rp)
        tr.append_point(b"L", Lp)
        tr.append_point(b"R", Rp)
        u = tr.challenge_scalar(b"u")
        u_inv = inv(u)
        a = [(aL[i] * u + u_inv * aR[i]) % L for i in range(n)]
        b = [(bL[i] * u_inv + u * bR[i]) % L for i in range(n)]
        if first:
            g_lo = sc_join([u_inv * G_factors[i] % L for i in range(n)])
            g_hi = sc_join([u * G_factors[n + i] % L for i in range(n)])
            h_lo = sc_join([u * H_factors[i] % L for i in range(n)])
            h_hi = sc_join([u_inv * H_factors[n + i] % L for i in range(n)])
            G = orc.fold_points(G, n, g_lo, g_hi, shared=False)
            H = orc.fold_points(H, n, h_lo, h_hi, shared=False)
        else:
            G = orc.fold_points(G, n, sc(u_inv), sc(u), shared=True)
            H = orc.fold_points(H, n, sc(u), sc(u_inv), shared=True)
        first = False
    return InnerProductProof(L_vec, R_vec, a[0], b[0])


def ipp_verification_scalars(ipp, n, tr):
    lg_n = len(ipp.L_vec)
    if lg_n >= 32 or n != (1 << lg_n):
        raise VerificationError("ipp size")
    tr.innerproduct_domain_sep(n)
    ch = []
    for Lp, Rp in zip(ipp.L_vec, ipp.R_vec):
        tr.validate_and_append_point(b"L", Lp)
        tr.validate_and_append_point(b"R", Rp)
        ch.append(tr.challenge_scalar(b"u"))
    ch_inv = [inv(c) for c in ch]
    allinv = 1
    for c in ch_inv:
        allinv = allinv * c % L
    ch_sq = [c * c % L for c in ch]
    ch_inv_sq = [c * c % L for c in ch_inv]
    s = [allinv]
    for i in range(1, n):
        lg_i = i.bit_length() - 1
        k = 1 << lg_i
        s.append(s[i - k] * ch_sq[(lg_n - 1) - lg_i] % L)
    return ch_sq, ch_inv_sq, s


class Verifier(_CS):
    def __init__(self, orc, transcript):
        self.orc, self.transcript = orc, transcript
        transcript.r1cs_domain_sep()
        self.constraints, self.num_vars, self.V = [], 0, []
        self.deferred = []

    def commit(self, V32):
        i = len(self.V)
        self.V.append(V32)
        self.transcript.append_point(b"V", V32)
        return Variable("C", i)

    def multiply(self, left, right):
        left, right = LinearCombination.of(left), LinearCombination.of(right)
        i = self.num_vars
        self.num_vars += 1
        lv, rv, ov = Variable("L", i), Variable("R", i), Variable("O", i)
        self.constrain(left - lv)
        self.constrain(right - rv)
        return lv, rv, ov

    def allocate_multiplier(self, assignment):
        i = self.num_vars
        self.num_vars += 1
        return Variable("L", i), Variable("R", i), Variable("O", i)

    def verify(self, proof, pc_gens, bp_gens, external_rng32=bytes(32)):
        orc, tr = self.orc, self.transcript
        tr.append_u64(b"m", len(self.V))
        n1 = self.num_vars
        tr.validate_and_append_point(b"A_I1", proof.A_I1)
        tr.validate_and_append_point(b"A_O1", proof.A_O1)
        tr.validate_and_append_point(b"S1", proof.S1)
        self._create_randomized_constraints()
        n = self.num_vars
        n2 = n - n1
        padded_n = 1 if n == 0 else 1 << (n - 1).bit_length()
        pad = padded_n - n
        if bp_gens.gens_capacity < padded_n:
            raise InvalidGeneratorsLength()
        tr.append_point(b"A_I2", proof.A_I2)
        tr.append_point(b"A_O2", proof.A_O2)
        tr.append_point(b"S2", proof.S2)
        y = tr.challenge_scalar(b"y")
        z = tr.challenge_scalar(b"z")
        for lab, T in ((b"T_1", proof.T_1), (b"T_3", proof.T_3), (b"T_4", proof.T_4), (b"T_5", proof.T_5), (b"T_6", proof.T_6)):
            tr.validate_and_append_point(lab, T)
        u = tr.challenge_scalar(b"u")
        x = tr.challenge_scalar(b"x")
        tr.append_scalar(b"t_x", proof.t_x)
        tr.append_scalar(b"t_x_blinding", proof.t_x_blinding)
        tr.append_scalar(b"e_blinding", proof.e_blinding)
        w = tr.challenge_scalar(b"w")
        wL, wR, wO, wV, wc = self._flatten(z, n, len(self.V), True)
        u_sq, u_inv_sq, s = ipp_verification_scalars(proof.ipp, padded_n, tr)
        a, b = proof.ipp.a, proof.ipp.b
        y_inv = inv(y)
        y_inv_vec = [1] * padded_n
        for i in range(1, padded_n):
            y_inv_vec[i] = y_inv_vec[i - 1] * y_inv % L
        yneg_wR = [wR[i] * y_inv_vec[i] % L for i in range(n)] + [0] * pad
        delta = sum(yneg_wR[i] * wL[i] for i in range(n)) % L
        u_for = [1] * n1 + [u] * (n2 + pad)
        g_scalars = [u_for[i] * (x * yneg_wR[i] - a * s[i]) % L for i in range(padded_n)]
        wLp, wOp = wL + [0] * pad, wO + [0] * pad
        h_scalars = [u_for[i] * (y_inv_vec[i] * (x * wLp[i] + wOp[i] - b * s[padded_n - 1 - i]) - 1) % L for i in range(padded_n)]
        r = tr.build_rng([], external_rng32).scalar()
        xx = x * x % L
        rxx = r * xx % L
        xxx = x * xx % L
        T_scalars = [r * x, rxx * x, rxx * xx, rxx * xxx, rxx * xx % L * xx]
        scalars = ([x, xx, xxx, u * x, u * xx, u * xxx] + [wVi * rxx for wVi in wV] + T_scalars +
                   [w * (proof.t_x - a * b) + r * (xx * (wc + delta) - proof.t_x), -proof.e_blinding - r * proof.t_x_blinding] +
                   g_scalars + h_scalars + u_sq + u_inv_sq)
        dyn = ([proof.A_I1, proof.A_O1, proof.S1, proof.A_I2, proof.A_O2, proof.S2] + self.V +
               [proof.T_1, proof.T_3, proof.T_4, proof.T_5, proof.T_6])
        arr1, rc = orc.decode(b"".join(dyn))
        if arr1 is None:
            raise VerificationError("decompress")
        arr2, rc = orc.decode(b"".join(proof.ipp.L_vec + proof.ipp.R_vec))
        if arr2 is None:
            raise VerificationError("decompress")
        pts, cnt = orc.concat((arr1, len(dyn)), (pc_gens.B, 1), (pc_gens.B_blinding, 1),
                              (orc.slice(bp_gens.G, 0, padded_n), padded_n), (orc.slice(bp_gens.H, 0, padded_n), padded_n),
                              (arr2, 2 * len(proof.ipp.L_vec)))
        assert cnt == len(scalars)
        if orc.msm(sc_join(scalars), pts, cnt, binding.SCHED_VARTIME) != bytes(32):
            raise VerificationError("mega check")
        return True
