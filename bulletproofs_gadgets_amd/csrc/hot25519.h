// hot25519.h -- device-only field / group arithmetic of the MSM's hot kernels (gfx950).
//
// Same representation and results as fe25519.h / ge25519.h (eight 32-bit limbs, value in
// [0, 2^256), 2^256 = 38), written for how gfx950 issues it:
//
//  * v_mad_u64_u32 takes a 64-bit addend in an even-aligned VGPR pair.  The row-wise product
//    adds one 32-bit accumulator word to every partial product, i.e. the pair (word, 0).  Left
//    to itself the compiler re-creates that zero with a v_mov_b32 in front of almost every
//    multiply-add: 425 v_mov per mixed addition beside 504 v_mad_u64_u32 (round-1 disassembly).
//    Here every accumulator word k owns a partner register Z.z[k] that holds zero for the whole
//    kernel and that the compiler cannot see through (it is written by an asm statement), and the
//    multiply-add itself is an asm statement so that it cannot be re-associated into
//    (a*b + word) + (zero << 32).  The register coalescer then keeps (word k, Z.z[k]) in one
//    pair and writes each new word straight into its low half: no v_mov.
//  * Carry ripples that almost never run (adding 38 * (a carry bit) or 38 * (a 6-bit top word)
//    to limb 0 overflows it with probability < 2^-21) sit behind a branch on the carry mask
//    instead of eight v_addc_co_u32 on every call.
//  * A signed Pippenger digit negates the table point.  The table record holds
//    ((y+x)/2, (y-x)/2, dxy, -dxy), so the negated point is the same record read in another order:
//    an address choice instead of 48 v_cndmask_b32 per addition.  (Halved: see ge25519.h -- the
//    addition then uses Z where the usual form uses 2Z.)
//
// Replaces, for Prover::prove / Verifier::verify's multiscalar multiplications, what
// curve25519-dalek's FieldElement / EdwardsPoint do under Straus / Pippenger
// (ref call sites: tests/edwards_point.rs:47,79; Cargo.toml:12).
#pragma once
#include "ge25519.h"

#if defined(__HIPCC__)
namespace bpg {

// one opaque zero per accumulator word of the 512-bit product (see above)
struct fe_zeros {
    uint32_t z[16];
};
__device__ __forceinline__ void fe_zeros_init(fe_zeros &Z) {
#pragma unroll
    for (int k = 0; k < 16; k++) asm volatile("v_mov_b32 %0, 0" : "=v"(Z.z[k]));
}

// d = a * b + c, c a 64-bit register pair; the carry-out (never set: a * b + a 32-bit word fits 64 bits) goes to a
// scratch SGPR pair.  An asm statement, so that a * b + (word, zero) cannot be re-associated into
// (a * b + word) + (zero << 32).  (Laundering the pair through an empty asm and leaving the multiply-add to the
// compiler was tried: the tied operand costs a v_mov_b64 per pair.)
__device__ __forceinline__ uint64_t mad64(uint32_t a, uint32_t b, uint64_t c) {
    uint64_t d, junk;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(junk) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint64_t mul64(uint32_t a, uint32_t b) {
    uint64_t d, junk;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(junk) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint64_t zpair(uint32_t lo, uint32_t zero) { return ((uint64_t)zero << 32) | lo; }

// Marks a block as not worth speculating: the compiler keeps the branch around it.
#define BPG_RARE_PATH() asm volatile("" ::: "memory")

// r = a + b (mod p), weakly reduced
__device__ __forceinline__ void fez_add(fe &r, const fe &a, const fe &b) {
    uint32_t c = 0, t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = addc32(a.v[i], b.v[i], c, c);
    uint32_t c2 = 0;
    t[0] = addc32(t[0], c ? 38u : 0u, 0, c2);
    if (__builtin_expect(c2 != 0, 0)) {
        // t[0] wrapped: the sum was within 38 of 2^256.  Ripple; it cannot carry out again
        // (the true sum is below 2^257, so after one fold the value is below 2^256 + 38).
        BPG_RARE_PATH();
        uint32_t k = 1;
#pragma unroll
        for (int i = 1; i < 8; i++) t[i] = addc32(t[i], 0, k, k);
        t[0] += k ? 38u : 0u;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
}

// r = a - b (mod p), weakly reduced
__device__ __forceinline__ void fez_sub(fe &r, const fe &a, const fe &b) {
    uint32_t c = 0, t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = subb32(a.v[i], b.v[i], c, c);
    uint32_t c2 = 0;
    t[0] = subb32(t[0], c ? 38u : 0u, 0, c2);
    if (__builtin_expect(c2 != 0, 0)) {
        BPG_RARE_PATH();
        uint32_t k = 1;
#pragma unroll
        for (int i = 1; i < 8; i++) t[i] = subb32(t[i], 0, k, k);
        t[0] -= k ? 38u : 0u;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
}

// acc[0..15] (512-bit) -> r (weakly reduced)
__device__ __forceinline__ void fez_reduce512(fe &r, const uint32_t acc[16], const fe_zeros &Z) {
    uint32_t lo[8], hi[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint64_t t = mad64(acc[8 + j], 38u, zpair(acc[j], Z.z[j]));
        lo[j] = (uint32_t)t;
        hi[j] = (uint32_t)(t >> 32);
    }
    uint32_t w[8], c = 0;
    w[0] = lo[0];
#pragma unroll
    for (int j = 1; j < 8; j++) w[j] = addc32(lo[j], hi[j - 1], c, c);
    const uint32_t top = hi[7] + c;           // <= 38
    uint32_t c2 = 0;
    w[0] = addc32(w[0], top * 38u, 0, c2);
    if (__builtin_expect(c2 != 0, 0)) {
        BPG_RARE_PATH();
        uint32_t k = 1;
#pragma unroll
        for (int j = 1; j < 8; j++) w[j] = addc32(w[j], 0, k, k);
        w[0] += k ? 38u : 0u;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) r.v[j] = w[j];
}

// r = a * b (mod p): 64 + 8 v_mad_u64_u32, one v_addc_co_u32 per partial product, nothing else
__device__ __forceinline__ void fez_mul(fe &r, const fe &a, const fe &b, const fe_zeros &Z) {
    uint32_t acc[16];
    {
        uint32_t lo[8], hi[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint64_t t = mul64(a.v[0], b.v[j]);
            lo[j] = (uint32_t)t;
            hi[j] = (uint32_t)(t >> 32);
        }
        uint32_t c = 0;
        acc[0] = lo[0];
#pragma unroll
        for (int j = 1; j < 8; j++) acc[j] = addc32(lo[j], hi[j - 1], c, c);
        acc[8] = addc32(hi[7], 0, c, c);
    }
#pragma unroll
    for (int i = 1; i < 8; i++) {
        uint32_t lo[8], hi[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint64_t t = mad64(a.v[i], b.v[j], zpair(acc[i + j], Z.z[i + j]));
            lo[j] = (uint32_t)t;
            hi[j] = (uint32_t)(t >> 32);
        }
        uint32_t c = 0;
        acc[i] = lo[0];
#pragma unroll
        for (int j = 1; j < 8; j++) acc[i + j] = addc32(lo[j], hi[j - 1], c, c);
        acc[i + 8] = addc32(hi[7], 0, c, c);
    }
    fez_reduce512(r, acc, Z);
}

// The three field elements of a table record as a mixed addition consumes them: ((y+x)/2, (y-x)/2, dxy) of +P, or
// for a negative digit ((y-x)/2, (y+x)/2, -dxy), i.e. the record of -P.
struct ge_aniels_signed {
    fe A, B, C;          // multiply (Y+X), (Y-X), T
};

// r = p + q, q as read by the sign; 7M, 7 add/sub, no selects
__device__ __forceinline__ void gez_madd(ge &r, const ge &p, const ge_aniels_signed &q, const fe_zeros &Z) {
    fe a, b, c, e, f, g, h, t;
    fez_sub(t, p.Y, p.X);
    fez_add(h, p.Y, p.X);
    fez_mul(a, t, q.B, Z);
    fez_mul(b, h, q.A, Z);
    fez_mul(c, p.T, q.C, Z);
    fez_sub(e, b, a);
    fez_add(h, b, a);
    fez_sub(f, p.Z, c);
    fez_add(g, p.Z, c);
    fez_mul(r.X, e, f, Z);
    fez_mul(r.Y, g, h, Z);
    fez_mul(r.T, e, h, Z);
    fez_mul(r.Z, f, g, Z);
}

// The same addition as ONE hand-scheduled instruction stream (tools/gen_madd_asm.py): fixed (word, zero) register
// pairs, three carry chains in flight, no s_nop, no spills.  p += q; q is overwritten.
#include "madd_gfx950.inc"

// r = identity + q = q: x = A - B, y = A + B, Z = 1, T = xy -- one multiplication instead of seven
__device__ __forceinline__ void gez_from_aniels(ge &r, const ge_aniels_signed &q) {
    fe_sub(r.X, q.A, q.B);
    fe_add(r.Y, q.A, q.B);
    fe_set1(r.Z);
    fe_mul(r.T, r.X, r.Y);
}

// r = p + q (both extended), 9M
__device__ __forceinline__ void gez_add(ge &r, const ge &p, const ge &q, const fe_zeros &Z) {
    fe a, b, c, d, e, f, g, h, t, u;
    fez_sub(t, p.Y, p.X);
    fez_sub(u, q.Y, q.X);
    fez_mul(a, t, u, Z);
    fez_add(t, p.Y, p.X);
    fez_add(u, q.Y, q.X);
    fez_mul(b, t, u, Z);
    fez_mul(c, p.T, q.T, Z);
    fez_mul(c, c, fe_d2(), Z);
    fez_mul(d, p.Z, q.Z, Z);
    fez_add(d, d, d);
    fez_sub(e, b, a);
    fez_sub(f, d, c);
    fez_add(g, d, c);
    fez_add(h, b, a);
    fez_mul(r.X, e, f, Z);
    fez_mul(r.Y, g, h, Z);
    fez_mul(r.T, e, h, Z);
    fez_mul(r.Z, f, g, Z);
}

}  // namespace bpg
#endif
