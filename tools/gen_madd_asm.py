#!/usr/bin/env python3
"""Generator of bulletproofs_gadgets_amd/csrc/madd_gfx950.inc: the mixed addition of the MSM's hot kernel
(extended point += affine-Niels table point, 7 field multiplications mod 2^255 - 19) as ONE hand-scheduled
gfx950 instruction stream.

Why not leave it to the compiler (measured, rounds 1 and 2):
  * v_mad_u64_u32 adds a 64-bit VGPR pair.  The row-wise product adds a 32-bit accumulator word, i.e. the pair
    (word, 0): the compiler re-creates the zero with a v_mov_b32 per multiply-add (425 per addition in round 1).
    Here every accumulator word lives in the low half of a fixed pair whose high half is zeroed once per addition.
  * A VALU instruction that reads a carry (SGPR pair / VCC) written by a VALU instruction needs two other
    instructions in between on gfx950.  The compiler fills what it cannot schedule around with s_nop (187-668 per
    addition depending on the source form).  Here three carry chains are kept in flight (rows of the product are
    software-pipelined: the additions of row i run between the multiply-adds of rows i+1 and i+2), so the stream
    has (almost) no s_nop, and none of the compiler's 72 bytes of scratch spills.
  * Carry ripples that almost never run sit behind a scalar branch.

What this script does:
  1. builds the operation list of one mixed addition with registers assigned by hand (operands of the asm
     statement for the point and the table entry, a fixed block of scratch VGPRs / SGPRs declared as clobbers);
  2. list-schedules it: any order that respects every read-after-write, write-after-read and write-after-write
     dependency on every register is legal; among those it looks for one without carry-hazard stalls;
  3. runs the scheduled stream in an emulator of the dozen instructions used, on random and edge-case inputs,
     against Python integers (the result must be congruent mod p, below 2^256, and a valid extended point), and
     checks the two-wait-state rule on every carry read;
  4. writes the .inc file (asm text + operand lists).  The build does not run this script; the .inc is committed.

Field representation: eight 32-bit limbs, value in [0, 2^256), 2^256 = 38 (mod p) -- the same as fe25519.h.
Usage: python3 tools/gen_madd_asm.py [--check-only]
"""
import random
import sys
from collections import defaultdict

P25519 = 2**255 - 19
M32 = 0xFFFFFFFF

# ---------------------------------------------------------------------------------------------- register map
SB = 72                                   # first scratch VGPR; the compiler keeps v0..v71 for itself
def Plo(k): return "v%d" % (SB + 2 * k)           # accumulator word k of the 512-bit product
def Phi(k): return "v%d" % (SB + 2 * k + 1)       # its partner: zero whenever the pair is an addend
def Ppair(k): return "v[%d:%d]" % (SB + 2 * k, SB + 2 * k + 1)
def Tlo(j): return "v%d" % (SB + 32 + 2 * j)      # product a_i * b_j (+ word), column j
def Thi(j): return "v%d" % (SB + 32 + 2 * j + 1)
def Tpair(j): return "v[%d:%d]" % (SB + 32 + 2 * j, SB + 32 + 2 * j + 1)
SPARE = ["v%d" % (SB + 48 + i) for i in range(8)]  # one field element (h, then b, then h again)
N_SCRATCH_V = 56
CARRY = ["vcc", "s[96:97]", "s[98:99]"]          # carry chains in flight
RARE = "s[94:95]"                                 # carry out of a limb-0 fold: tested by the scalar unit
JUNK = "s[100:101]"                               # carry-out of v_mad_u64_u32 (never set)
CLOBBER_S = ["s94", "s95", "s96", "s97", "s98", "s99", "s100", "s101", "vcc", "scc"]

# operands of the asm statement: %0..%31 = X, Y, Z, T of the accumulator (in/out); %32..%55 = A, B, C of the table
# entry as the sign of the digit selected them (in; overwritten: the block uses them as scratch once consumed)
def opnd(i): return "%%%d" % i
X = [opnd(i) for i in range(0, 8)]
Y = [opnd(i) for i in range(8, 16)]
Z = [opnd(i) for i in range(16, 24)]
T = [opnd(i) for i in range(24, 32)]
QA = [opnd(i) for i in range(32, 40)]
QB = [opnd(i) for i in range(40, 48)]
QC = [opnd(i) for i in range(48, 56)]


def pair_regs(p):
    """'v[72:73]' -> ('v72', 'v73')"""
    a, b = p[2:-1].split(":")
    return "v" + a, "v" + b


class Op:
    """One instruction, or one atomic group of instructions (a rarely taken ripple with its branch)."""
    __slots__ = ("text", "reads", "writes", "carry_in", "carry_out", "kind", "sem", "n_instr")

    def __init__(self, text, reads, writes, sem, carry_in=None, carry_out=None, kind="valu", n_instr=1):
        self.text, self.sem, self.kind, self.n_instr = text, sem, kind, n_instr
        self.reads, self.writes = set(reads), set(writes)
        self.carry_in, self.carry_out = carry_in, carry_out
        if carry_in:
            self.reads.add(carry_in)
        if carry_out and carry_out != JUNK:
            self.writes.add(carry_out)


# ---------------------------------------------------------------------------------------------- op constructors
def is_reg(s):
    return isinstance(s, str)


def val(st, s):
    return st[s] if is_reg(s) else s


def v_mov(d, s):
    def sem(st):
        st[d] = val(st, s)
    return Op("v_mov_b32 %s, %s" % (d, s), [s] if is_reg(s) else [], [d], sem)


def v_mad64(dpair, a, b, addend):
    """dpair = a * b + addend; addend a pair or 0; b may be an inline constant"""
    dl, dh = pair_regs(dpair)
    reads = [r for r in (a, b) if is_reg(r)]
    if addend != 0:
        al, ah = pair_regs(addend)
        reads += [al, ah]

    def sem(st):
        add = 0 if addend == 0 else (st[ah] << 32 | st[al])
        r = val(st, a) * val(st, b) + add
        assert r < 1 << 64, "v_mad_u64_u32 overflow: the zero partner is not zero"
        st[dl], st[dh] = r & M32, r >> 32
    return Op("v_mad_u64_u32 %s, %s, %s, %s, %s" % (dpair, JUNK, a, b, addend), reads, [dl, dh], sem, carry_out=JUNK)


def v_add_co(d, cout, a, b, cin=None, sub=False):
    """d = a +/- b +/- cin, carry / borrow to cout"""
    name = ("v_subb_co_u32" if cin else "v_sub_co_u32") if sub else ("v_addc_co_u32" if cin else "v_add_co_u32")
    text = "%s %s, %s, %s, %s" % (name, d, cout, a, b) + (", %s" % cin if cin else "")

    def sem(st):
        c = st[cin] if cin else 0
        r = val(st, a) - val(st, b) - c if sub else val(st, a) + val(st, b) + c
        st[d] = r & M32
        st[cout] = 1 if (r < 0 or r > M32) else 0
    return Op(text, [r for r in (a, b) if is_reg(r)], [d], sem, carry_in=cin, carry_out=cout)


def v_add_nc(d, a, b, sub=False):
    def sem(st):
        st[d] = (val(st, a) - val(st, b) if sub else val(st, a) + val(st, b)) & M32
    return Op("%s %s, %s, %s" % ("v_sub_u32" if sub else "v_add_u32", d, a, b), [r for r in (a, b) if is_reg(r)], [d], sem)


def v_carry_mask(d, c):
    """d = all ones if the carry (borrow) c is set, else 0: 0 - 0 - c.  (v_cndmask_b32 with an SGPR mask measured
    23 cycles per wave on gfx950, tools/irate.hip, against 4 for this; the v_and_b32 that follows is a 2-cycle op.)"""
    def sem(st):
        st[d] = M32 if st[c] else 0
    return Op("v_subb_co_u32 %s, %s, 0, 0, %s" % (d, c, c), [], [d], sem, carry_in=c, carry_out=c)


def v_and38(d):
    def sem(st):
        st[d] &= 38
    return Op("v_and_b32 %s, 38, %s" % (d, d), [d], [d], sem)


def v_mul38(d, a):
    def sem(st):
        assert st[a] < 1 << 24
        st[d] = st[a] * 38 & M32
    return Op("v_mul_u32_u24 %s, 38, %s" % (d, a), [a], [d], sem)


_label = [0]


def rare_ripple(limbs, k, sub=False):
    """The carry (borrow) out of limb 0's fold is in RARE.  If it is set in any lane: ripple it through limbs 1..7 and
    fold the carry out of limb 7 (times 38) into limb 0 -- a no-op for the lanes whose bit is clear.  One atomic op
    for the scheduler; the two-wait-state rule is kept inside with s_nop (this path runs once in ~10^5 calls)."""
    _label[0] += 1
    lab = "Lbpg%d_%%=" % _label[0]
    addc = "v_subb_co_u32" if sub else "v_addc_co_u32"
    lines = ["s_cmp_lg_u64 %s, 0" % RARE, "s_cbranch_scc0 %s" % lab, "s_nop 1"]
    for i in range(1, 8):
        lines += ["%s %s, %s, %s, 0, %s" % (addc, limbs[i], RARE, limbs[i], RARE), "s_nop 1"]
    lines += ["v_cndmask_b32 %s, 0, 38, %s" % (k, RARE), "%s %s, %s, %s" % ("v_sub_u32" if sub else "v_add_u32", limbs[0], limbs[0], k),
              "%s:" % lab]

    def sem(st):
        if not st[RARE] and not st.get("__force_rare__"):
            return
        c = st[RARE]
        for i in range(1, 8):
            r = st[limbs[i]] - c if sub else st[limbs[i]] + c
            st[limbs[i]] = r & M32
            c = 1 if (r < 0 or r > M32) else 0
        st[k] = 38 if c else 0
        st[limbs[0]] = (st[limbs[0]] - st[k] if sub else st[limbs[0]] + st[k]) & M32
        st[RARE] = c
    # the scalar compare reads RARE right after the VALU wrote it: interlocked by the hardware (the compiler emits
    # v_cmp ... s[6:7] / s_mov_b64 vcc, s[6:7] back to back); kind 'salu' exempts the op from the VALU-read rule
    return Op("\n".join(lines), list(limbs) + [RARE], list(limbs) + [k, RARE, "scc"], sem, kind="salu", n_instr=len(lines) - 1)


# ---------------------------------------------------------------------------------------------- field operations
def fe_addsub(ops, dst, a, b, carry, k, sub=False):
    """dst = a +/- b mod p (weakly reduced); k = a scratch VGPR"""
    ops.append(v_add_co(dst[0], carry, a[0], b[0], sub=sub))
    for i in range(1, 8):
        ops.append(v_add_co(dst[i], carry, a[i], b[i], cin=carry, sub=sub))
    ops.append(v_carry_mask(k, carry))
    ops.append(v_and38(k))
    ops.append(v_add_co(dst[0], RARE, dst[0], k, sub=sub))
    ops.append(rare_ripple(dst, k, sub=sub))


def fe_mul(ops, dst, a, b, k):
    """dst = a * b mod p.  Row i: T[j] = a_i * b_j + (word i+j, 0) for j = 0..7 (column 0 accumulates in place in
    P[i], whose low half is then word i of the product), then one carry chain adds the columns up:
    word i+j = lo(T[j]) + hi(T[j-1]) + carry.  Rows alternate between the carry registers so that the scheduler can
    overlap the chain of row i with the multiply-adds of the rows after it."""
    for i in range(8):
        c = CARRY[i % 3]
        a_i = a[i]
        ops.append(v_mad64(Ppair(i), a_i, b[0], 0 if i == 0 else Ppair(i)))
        for j in range(1, 8):
            ops.append(v_mad64(Tpair(j), a_i, b[j], 0 if i == 0 else Ppair(i + j)))
        ops.append(v_add_co(Plo(i + 1), c, Tlo(1), Phi(i)))
        ops.append(v_mov(Phi(i), 0))                         # P[i] is an addend again in the reduction
        for j in range(2, 8):
            ops.append(v_add_co(Plo(i + j), c, Tlo(j), Thi(j - 1), cin=c))
        ops.append(v_add_co(Plo(i + 8), c, 0, Thi(7), cin=c))
    # 512 -> 256 bits: U_j = word(8+j) * 38 + (word j, 0)
    c = CARRY[2]
    for j in range(8):
        ops.append(v_mad64(Tpair(j), Plo(8 + j), 38, Ppair(j)))
    ops.append(v_add_co(dst[1], c, Tlo(1), Thi(0)))
    for j in range(2, 8):
        ops.append(v_add_co(dst[j], c, Tlo(j), Thi(j - 1), cin=c))
    ops.append(v_add_co(k, c, 0, Thi(7), cin=c))             # top word <= 38
    ops.append(v_mul38(k, k))
    ops.append(v_add_co(dst[0], RARE, Tlo(0), k))
    ops.append(rare_ripple(dst, k))


def build_madd():
    """acc (X, Y, Z, T) += (A, B, C) = ((y+x)/2, (y-x)/2, dxy) of the table point (or of its negative): the
    add-2008-hwcd-3 addition with a precomputed point, every product halved (so Z stands where 2Z would).
    Where each value lives:
         h0 = Y+X -> SPARE     t = Y-X -> X
         a = t*B -> Y          b = h0*A -> SPARE    c = T*C -> T
         e = b-a -> QA         h = b+a -> SPARE     f = Z-c -> QB     g = Z+c -> QC
         X = e*f   Y = g*h   T = e*h   Z = f*g      (the accumulator's registers are free by then)"""
    ops = []
    for k in range(16):
        ops.append(v_mov(Phi(k), 0))
    K = [Tlo(0), Tlo(1), Tlo(2), Tlo(3)]                      # scratch words of the add / sub folds (T is idle there)
    fe_addsub(ops, SPARE, Y, X, CARRY[0], K[0])
    fe_addsub(ops, X, Y, X, CARRY[1], K[1], sub=True)
    fe_mul(ops, Y, X, QB, Thi(0))
    fe_mul(ops, SPARE, SPARE, QA, Thi(0))
    fe_mul(ops, T, T, QC, Thi(0))
    fe_addsub(ops, QA, SPARE, Y, CARRY[0], K[0], sub=True)
    fe_addsub(ops, SPARE, SPARE, Y, CARRY[1], K[1])
    fe_addsub(ops, QB, Z, T, CARRY[2], K[2], sub=True)
    fe_addsub(ops, QC, Z, T, CARRY[0], K[3])
    fe_mul(ops, X, QA, QB, Thi(0))
    fe_mul(ops, Y, QC, SPARE, Thi(0))
    fe_mul(ops, T, QA, SPARE, Thi(0))
    fe_mul(ops, Z, QB, QC, Thi(0))
    return ops


# ---------------------------------------------------------------------------------------------- scheduler
HAZARD_GAP = 2          # instructions between a VALU write of a carry register and a VALU read of it


def schedule(ops, window=48):
    """List scheduling.  Dependencies come from the register sets in program order (RAW, WAR, WAW).  At every slot
    the ready op with the longest remaining dependency chain is issued, unless its carry input was written fewer
    than HAZARD_GAP slots ago; if nothing can issue, an s_nop fills the slot.  `window` bounds how far ahead of the
    oldest unissued op the scheduler looks (register lifetimes stay short, the stream stays readable)."""
    n = len(ops)
    succ, npred = defaultdict(set), [0] * n
    last_write, readers = {}, defaultdict(list)
    for i, op in enumerate(ops):
        deps = set()
        for r in op.reads:
            if r in last_write:
                deps.add(last_write[r])
        for r in op.writes:
            if r in last_write:
                deps.add(last_write[r])
            deps.update(readers[r])
        deps.discard(i)
        for d in deps:
            if i not in succ[d]:
                succ[d].add(i)
                npred[i] += 1
        for r in op.writes:
            last_write[r] = i
            readers[r] = []
        for r in op.reads:
            readers[r].append(i)
    # priority: longest path to the end, weighted by issue cost
    cost = [4 if o.text.startswith("v_mad_u64") else 2 * o.n_instr for o in ops]
    prio = [0] * n
    for i in range(n - 1, -1, -1):
        prio[i] = cost[i] + max((prio[s] for s in succ[i]), default=0)
    done, out, slot = [False] * n, [], 0
    carry_written_at = {}
    ready = [i for i in range(n) if npred[i] == 0]
    oldest = 0
    nops = 0
    while len(out) - nops < n:
        while oldest < n and done[oldest]:
            oldest += 1
        best = None
        for i in ready:
            if i > oldest + window:
                continue
            op = ops[i]
            if op.kind == "valu" and op.carry_in and slot - carry_written_at.get(op.carry_in, -99) - 1 < HAZARD_GAP:
                continue
            if best is None or prio[i] > prio[best] or (prio[i] == prio[best] and i < best):
                best = i
        if best is None:
            out.append(None)                                 # s_nop 0
            nops += 1
            slot += 1
            continue
        op = ops[best]
        ready.remove(best)
        done[best] = True
        out.append(best)
        if op.kind == "valu" and op.carry_out and op.carry_out != JUNK:
            carry_written_at[op.carry_out] = slot
        if op.kind == "salu":
            # the group's own VALU writes of RARE are followed by >= 2 instructions inside the group
            carry_written_at[RARE] = slot - 99
        slot += op.n_instr
        for s in succ[best]:
            npred[s] -= 1
            if npred[s] == 0:
                ready.append(s)
    return out, nops


def check_hazards(ops, order):
    """independent check of the final stream: distance between a VALU carry write and the next VALU read of it"""
    slot, written = 0, {}
    for idx in order:
        if idx is None:
            slot += 1
            continue
        op = ops[idx]
        if op.kind == "valu":
            if op.carry_in:
                gap = slot - written.get(op.carry_in, -99) - 1
                assert gap >= HAZARD_GAP, "carry hazard: %s (gap %d)" % (op.text, gap)
            if op.carry_out and op.carry_out != JUNK:
                written[op.carry_out] = slot
        else:
            written[RARE] = -99
        slot += op.n_instr


# ---------------------------------------------------------------------------------------------- emulator / self-test
def limbs_of(x):
    return [(x >> (32 * i)) & M32 for i in range(8)]


def int_of(st, regs):
    return sum(st[r] << (32 * i) for i, r in enumerate(regs))


def run(ops, order, acc, q, force_rare=False):
    st = defaultdict(lambda: 0xDEADBEEF)        # scratch starts dirty: the block must not rely on anything
    for c in CARRY + [RARE]:
        st[c] = 1
    for regs, v in zip((X, Y, Z, T, QA, QB, QC), list(acc) + list(q)):
        for r, w in zip(regs, limbs_of(v)):
            st[r] = w
    if force_rare:
        st["__force_rare__"] = 1
    for idx in order:
        if idx is not None:
            ops[idx].sem(st)
    return [int_of(st, regs) for regs in (X, Y, Z, T)]


D2 = 2 * (-121665 * pow(121666, P25519 - 2, P25519)) % P25519


def ref_madd(acc, q):
    x, y, z, t = acc
    A, B, C = q
    a = (y - x) * B
    b = (y + x) * A
    c = t * C
    d = z
    e, h, f, g = b - a, b + a, d - c, d + c
    return [e * f % P25519, g * h % P25519, f * g % P25519, e * h % P25519]     # X, Y, Z, T


def self_test(ops, order, rounds=300):
    rnd = random.Random(2025)
    edge = [0, 1, 38, P25519 - 1, P25519, P25519 + 1, 2**255 - 1, 2**255, 2**256 - 1, 2**256 - 38, 2**256 - 39, 2**256 - 37,
            M32, M32 << 224, (2**256 - 1) ^ (M32 << 96), 19, 2**256 - 19]
    cases = []
    for _ in range(rounds):
        cases.append(([rnd.getrandbits(256) for _ in range(4)], [rnd.getrandbits(256) for _ in range(3)]))
    for _ in range(rounds):
        cases.append(([rnd.choice(edge) for _ in range(4)], [rnd.choice(edge) for _ in range(3)]))
    for _ in range(rounds):
        cases.append(([rnd.choice(edge + [rnd.getrandbits(256)]) for _ in range(4)], [rnd.choice(edge + [rnd.getrandbits(256)]) for _ in range(3)]))
    # inputs that make the folds wrap limb 0 (the rare ripples): a sum within 38 of 2^256, equal operands of a subtraction
    for _ in range(rounds):
        y = rnd.getrandbits(256)
        x = (2**256 - rnd.randrange(1, 60) - y) % 2**256
        cases.append(([x, y, rnd.choice([2**255 + rnd.randrange(40), 2**256 - 1 - rnd.randrange(40)]), rnd.getrandbits(256)],
                      [rnd.getrandbits(256) for _ in range(3)]))
    for acc, q in cases:
        want = ref_madd(acc, q)
        for force in (False, True):
            got = run(ops, order, acc, q, force_rare=force)
            gx, gy, gz, gt = got
            assert all(0 <= v < 2**256 for v in got)
            assert [gx % P25519, gy % P25519, gz % P25519, gt % P25519] == [want[0], want[1], want[2], want[3]], (acc, q, force)
    # a real point: identity + basepoint, then doubling-free chain; result must stay on the curve
    d = (-121665 * pow(121666, P25519 - 2, P25519)) % P25519
    by = 4 * pow(5, P25519 - 2, P25519) % P25519
    bx = 15112221349535400772501151409588531511454012693041857206046113283949847762202
    half = (P25519 + 1) // 2
    qb = [(by + bx) * half % P25519, (by - bx) * half % P25519, d * bx * by % P25519]
    acc = [0, 1, 1, 0]
    for _ in range(20):
        acc = run(ops, order, acc, qb)
        xx, yy, zz, tt = (v % P25519 for v in acc)
        assert (yy * yy - xx * xx - zz * zz - d * tt * tt) % P25519 == 0 and (xx * yy - zz * tt) % P25519 == 0
    return len(cases)


# ---------------------------------------------------------------------------------------------- output
def emit(ops, order, path):
    lines = []
    for idx in order:
        if idx is None:
            lines.append("s_nop 0")
        else:
            lines += ops[idx].text.split("\n")
    n_mad = sum(1 for l in lines if l.startswith("v_mad_u64"))
    n_nop = sum(1 for l in lines if l.startswith("s_nop"))
    n_valu = sum(1 for l in lines if l.startswith("v_"))
    body = "\n".join('        "%s\\n"' % l for l in lines)
    accs = ",\n          ".join(", ".join('"+v"(%s.v[%d])' % (f, i) for i in range(8)) for f in ("p.X", "p.Y", "p.Z", "p.T"))
    qs = ",\n          ".join(", ".join('"+v"(%s.v[%d])' % (f, i) for i in range(8)) for f in ("q.A", "q.B", "q.C"))
    clob = ", ".join('"v%d"' % (SB + i) for i in range(N_SCRATCH_V)) + ",\n          " + ", ".join('"%s"' % s for s in CLOBBER_S)
    with open(path, "w") as f:
        f.write("""// madd_gfx950.inc -- GENERATED by tools/gen_madd_asm.py (which also verified this stream in its emulator:
// results against Python integers on random and edge-case inputs, and the two-wait-state rule on every carry read).
// Do not edit by hand; edit the generator and run it again.
//
// p += q for an extended point p and a table entry q = (A, B, C) = ((y+x)/2, (y-x)/2, dxy) of +P or ((y-x)/2, (y+x)/2, -dxy) of -P.
// %d instructions: %d v_mad_u64_u32, %d other VALU, %d s_nop.  Scratch: v%d..v%d, s94..s101, vcc (clobbers).
// q is overwritten (its registers hold e, f, g once A, B, C have been consumed).
__device__ __forceinline__ void gez_madd_asm(ge &p, ge_aniels_signed &q) {
    asm volatile(
%s
        : %s,
          %s
        :
        : %s);
}
""" % (len(lines), n_mad, n_valu - n_mad, n_nop, SB, SB + N_SCRATCH_V - 1, body, accs, qs, clob))
    return len(lines), n_mad, n_valu - n_mad, n_nop


def main():
    ops = build_madd()
    order, nops = schedule(ops)
    check_hazards(ops, order)
    ncases = self_test(ops, order)
    # the unscheduled program order must give the same results (checks the dependency extraction, not the maths)
    assert run(ops, list(range(len(ops))), [5, 7, 11, 13], [17, 19, 23]) == run(ops, order, [5, 7, 11, 13], [17, 19, 23])
    print("ops %d, s_nop %d, %d emulator cases x 2 modes ok" % (len(ops), nops, ncases))
    if "--check-only" not in sys.argv:
        import os
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bulletproofs_gadgets_amd", "csrc", "madd_gfx950.inc")
        if "--out" in sys.argv:                  # tests/test_generated_stream.py: regenerate elsewhere and compare with the committed file
            path = sys.argv[sys.argv.index("--out") + 1]
        print("wrote %s: %d instructions, %d v_mad_u64_u32, %d other VALU, %d s_nop" % ((path,) + emit(ops, order, path)))


if __name__ == "__main__":
    main()
