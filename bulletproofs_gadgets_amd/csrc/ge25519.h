// ge25519.h -- edwards25519 / ristretto255 group operations for the MI355X
// product path (host + device).
//
// Point formats
//   ge        extended (X:Y:Z:T), the accumulator format (128 B)
//   ge_aniels halved affine Niels ((y+x)/2, (y-x)/2, dxy): the resident table
//             format (with -dxy in the fourth 32 bytes of the 128-byte record,
//             one cache line per point).  Against the usual (y+x, y-x, 2dxy) every
//             product of the mixed addition comes out halved, the sum a quarter of
//             the usual projective representative -- the same point -- and the
//             addition needs Z itself where the usual form needs 2Z: one field
//             addition fewer per point addition.
// Mixed addition ge + ge_aniels is 7 field multiplications; signed Pippenger
// digits negate by swapping the first two fields and negating the third.
//
// Replaces curve25519-dalek's EdwardsPoint / RistrettoPoint under
// Prover::prove / Verifier::verify / Prover::commit / BulletproofGens::new
// (ref call sites: tests/edwards_point.rs:47,79,90; src/gadgets/point/edwards_point.rs:267).
// Encodings follow RFC 9496 (ristretto255), which is what dalek implements.
#pragma once
#include "fe25519.h"

namespace bpg {

struct ge {
    fe X, Y, Z, T;
};
struct ge_aniels {
    fe hYpX, hYmX, Tdxy;      // (y+x)/2, (y-x)/2, d*x*y
};

#define BPG_FE_CONST(name, ...)                 \
    BPG_HD fe name() {                          \
        fe r = {{__VA_ARGS__}};                 \
        return r;                               \
    }
BPG_FE_CONST(fe_d, 0x135978a3u, 0x75eb4dcau, 0x4141d8abu, 0x00700a4du, 0x7779e898u, 0x8cc74079u, 0x2b6ffe73u, 0x52036ceeu)
BPG_FE_CONST(fe_d2, 0x26b2f159u, 0xebd69b94u, 0x8283b156u, 0x00e0149au, 0xeef3d130u, 0x198e80f2u, 0x56dffce7u, 0x2406d9dcu)
BPG_FE_CONST(fe_sqrtm1, 0x4a0ea0b0u, 0xc4ee1b27u, 0xad2fe478u, 0x2f431806u, 0x3dfbd7a7u, 0x2b4d0099u, 0x4fc1df0bu, 0x2b832480u)
BPG_FE_CONST(fe_sqrt_ad_m1, 0x497b2e1bu, 0x7e97f6a0u, 0x1b7854bdu, 0xaf9d8e0cu, 0x31f5d1fdu, 0x0f3cfcc9u, 0x2b8348acu, 0x376931bfu)
BPG_FE_CONST(fe_invsqrt_a_md, 0x805d40eau, 0x99c8fdaau, 0x5a4172beu, 0x9d2f1617u, 0xfe01d840u, 0x16c27b91u, 0xcfaffca2u, 0x786c8905u)
BPG_FE_CONST(fe_one_m_dsq, 0x945fc176u, 0xe27c09c1u, 0xcd5e350fu, 0x2c81a138u, 0xbe70dfe4u, 0x9994abddu, 0xb2b3e0d7u, 0x029072a8u)
BPG_FE_CONST(fe_d_m1_sq, 0x44ed4d20u, 0x31ad5aaau, 0xb01e1999u, 0xd29e4a2cu, 0x529b4eebu, 0x4cdcd32fu, 0xf66c2241u, 0x5968b37au)
BPG_FE_CONST(fe_bx, 0x8f25d51au, 0xc9562d60u, 0x9525a7b2u, 0x692cc760u, 0xfdd6dc5cu, 0xc0a4e231u, 0xcd6e53feu, 0x216936d3u)
BPG_FE_CONST(fe_by, 0x66666658u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u)
BPG_FE_CONST(fe_bt, 0xa5b7dda3u, 0x6dde8ab3u, 0x775152f5u, 0x20f09f80u, 0x64abe37du, 0x66ea4e8eu, 0xd78b7665u, 0x67875f0fu)

BPG_HD void ge_identity(ge &p) {
    fe_set0(p.X);
    fe_set1(p.Y);
    fe_set1(p.Z);
    fe_set0(p.T);
}
BPG_HD void ge_basepoint(ge &p) {
    p.X = fe_bx();
    p.Y = fe_by();
    fe_set1(p.Z);
    p.T = fe_bt();
}

// r = p + q (q halved affine Niels), 7M.  neg: r = p - q.
BPG_HD void ge_madd(ge &r, const ge &p, const ge_aniels &q, bool neg = false) {
    fe a, b, c, e, f, g, h, t, yp, ym, f0, g0;
    fe_sub(t, p.Y, p.X);
    fe_add(h, p.Y, p.X);
    // lane-uniform control flow: the sign only drives selects (v_cndmask)
#pragma unroll
    for (int i = 0; i < 8; i++) {
        yp.v[i] = neg ? q.hYmX.v[i] : q.hYpX.v[i];
        ym.v[i] = neg ? q.hYpX.v[i] : q.hYmX.v[i];
    }
    fe_mul(a, t, ym);
    fe_mul(b, h, yp);
    fe_mul(c, p.T, q.Tdxy);
    fe_sub(e, b, a);
    fe_add(h, b, a);
    fe_sub(f0, p.Z, c);
    fe_add(g0, p.Z, c);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        f.v[i] = neg ? g0.v[i] : f0.v[i];
        g.v[i] = neg ? f0.v[i] : g0.v[i];
    }
    fe_mul(r.X, e, f);
    fe_mul(r.Y, g, h);
    fe_mul(r.T, e, h);
    fe_mul(r.Z, f, g);
}

// r = p + q (both extended), 9M
BPG_HD void ge_add(ge &r, const ge &p, const ge &q) {
    fe a, b, c, d, e, f, g, h, t, u;
    fe_sub(t, p.Y, p.X);
    fe_sub(u, q.Y, q.X);
    fe_mul(a, t, u);
    fe_add(t, p.Y, p.X);
    fe_add(u, q.Y, q.X);
    fe_mul(b, t, u);
    fe_mul(c, p.T, q.T);
    fe_mul(c, c, fe_d2());
    fe_mul(d, p.Z, q.Z);
    fe_add(d, d, d);
    fe_sub(e, b, a);
    fe_sub(f, d, c);
    fe_add(g, d, c);
    fe_add(h, b, a);
    fe_mul(r.X, e, f);
    fe_mul(r.Y, g, h);
    fe_mul(r.T, e, h);
    fe_mul(r.Z, f, g);
}

#if defined(__HIPCC__)
// xor-butterfly exchange of an extended point inside groups of `width` lanes
__device__ __forceinline__ void ge_shfl_xor(ge &r, const ge &p, int mask, int width) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        r.X.v[i] = __shfl_xor(p.X.v[i], mask, width);
        r.Y.v[i] = __shfl_xor(p.Y.v[i], mask, width);
        r.Z.v[i] = __shfl_xor(p.Z.v[i], mask, width);
        r.T.v[i] = __shfl_xor(p.T.v[i], mask, width);
    }
}
#endif

BPG_HD void ge_neg(ge &r, const ge &p) {
    fe_neg(r.X, p.X);
    r.Y = p.Y;
    r.Z = p.Z;
    fe_neg(r.T, p.T);
}

// r = 2p, 4S + 4M (dbl-2008-hwcd with a = -1)
BPG_HD void ge_double(ge &r, const ge &p) {
    fe xx, yy, zz2, xpy, s, yypxx, yymxx, cx, ct;
    fe_sq(xx, p.X);
    fe_sq(yy, p.Y);
    fe_sq(zz2, p.Z);
    fe_add(zz2, zz2, zz2);
    fe_add(xpy, p.X, p.Y);
    fe_sq(s, xpy);
    fe_add(yypxx, yy, xx);
    fe_sub(yymxx, yy, xx);
    fe_sub(cx, s, yypxx);
    fe_sub(ct, zz2, yymxx);
    fe_mul(r.X, cx, ct);
    fe_mul(r.Y, yypxx, yymxx);
    fe_mul(r.Z, yymxx, ct);
    fe_mul(r.T, cx, yypxx);
}

// the same without T (a doubling that is followed by another doubling, or by a conversion to affine)
BPG_HD void ge_double_xyz(ge &r, const ge &p) {
    fe xx, yy, zz2, xpy, s, yypxx, yymxx, cx, ct;
    fe_sq(xx, p.X);
    fe_sq(yy, p.Y);
    fe_sq(zz2, p.Z);
    fe_add(zz2, zz2, zz2);
    fe_add(xpy, p.X, p.Y);
    fe_sq(s, xpy);
    fe_add(yypxx, yy, xx);
    fe_sub(yymxx, yy, xx);
    fe_sub(cx, s, yypxx);
    fe_sub(ct, zz2, yymxx);
    fe_mul(r.X, cx, ct);
    fe_mul(r.Y, yypxx, yymxx);
    fe_mul(r.Z, yymxx, ct);
}

BPG_FE_CONST(fe_half, 0xfffffff7u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu)   // (p+1)/2

// affine (x, y) -> halved affine Niels
BPG_HD void ge_affine_to_aniels(ge_aniels &r, const fe &x, const fe &y) {
    fe t, hx, hy;
    fe_mul(hx, x, fe_half());
    fe_mul(hy, y, fe_half());
    fe_add(r.hYpX, hy, hx);
    fe_sub(r.hYmX, hy, hx);
    fe_mul(t, x, y);
    fe_mul(r.Tdxy, t, fe_d());
}
// extended -> halved affine Niels given zinv = 1/Z
BPG_HD void ge_to_aniels(ge_aniels &r, const ge &p, const fe &zinv) {
    fe x, y;
    fe_mul(x, p.X, zinv);
    fe_mul(y, p.Y, zinv);
    ge_affine_to_aniels(r, x, y);
}
// halved affine Niels -> extended with Z = 1: x = A - B, y = A + B
BPG_HD void ge_from_aniels(ge &p, const ge_aniels &q) {
    fe_sub(p.X, q.hYpX, q.hYmX);
    fe_add(p.Y, q.hYpX, q.hYmX);
    fe_set1(p.Z);
    fe_mul(p.T, p.X, p.Y);
}
BPG_HD void ge_aniels_identity(ge_aniels &r) {
    r.hYpX = fe_half();
    r.hYmX = fe_half();
    fe_set0(r.Tdxy);
}

// ------------------------------ ristretto255 ------------------------------
// RFC 9496 4.2 SQRT_RATIO_M1 (dalek FieldElement::sqrt_ratio_i)
BPG_HD bool fe_sqrt_ratio_m1(fe &out, const fe &u, const fe &v) {
    fe v3, v7, r, check, t, neg_u, neg_u_i, rp;
    fe_sq(v3, v);
    fe_mul(v3, v3, v);
    fe_sq(v7, v3);
    fe_mul(v7, v7, v);
    fe_mul(t, u, v7);
    fe_pow22523(t, t);
    fe_mul(r, u, v3);
    fe_mul(r, r, t);
    fe_sq(check, r);
    fe_mul(check, check, v);
    fe_neg(neg_u, u);
    fe_mul(neg_u_i, neg_u, fe_sqrtm1());
    bool correct = fe_eq(check, u);
    bool flipped = fe_eq(check, neg_u);
    bool flipped_i = fe_eq(check, neg_u_i);
    fe_mul(rp, r, fe_sqrtm1());
    fe_cmov(r, rp, flipped | flipped_i);
    if (fe_isnegative(r)) fe_neg(r, r);
    out = r;
    return correct | flipped;
}

// CompressedRistretto::decompress -> affine (x, y); false if invalid.  in = 8 LE words.
BPG_HD bool ristretto_decode_affine_w(fe &x, fe &y, const uint32_t in[8]) {
    fe s, ss, u1, u2, u2sq, v, t, inv, den_x, den_y, one;
    uint32_t chk[8];
    fe_fromwords(s, in);
    fe_towords(chk, s);
    bool canon = true;
#pragma unroll
    for (int i = 0; i < 8; i++) canon &= (chk[i] == in[i]);
    if (!canon || (chk[0] & 1)) return false;
    fe_set1(one);
    fe_sq(ss, s);
    fe_sub(u1, one, ss);
    fe_add(u2, one, ss);
    fe_sq(u2sq, u2);
    fe_sq(t, u1);
    fe_mul(t, t, fe_d());
    fe_neg(t, t);
    fe_sub(v, t, u2sq);
    fe_mul(t, v, u2sq);
    bool ok = fe_sqrt_ratio_m1(inv, one, t);
    fe_mul(den_x, inv, u2);
    fe_mul(den_y, inv, den_x);
    fe_mul(den_y, den_y, v);
    fe_add(x, s, s);
    fe_mul(x, x, den_x);
    if (fe_isnegative(x)) fe_neg(x, x);
    fe_mul(y, u1, den_y);
    fe_mul(t, x, y);
    if (!ok || fe_isnegative(t) || fe_iszero(y)) return false;
    return true;
}

BPG_HD bool ristretto_decode_affine(fe &x, fe &y, const uint8_t in[32]) {
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; i++)
        w[i] = (uint32_t)in[4 * i] | ((uint32_t)in[4 * i + 1] << 8) | ((uint32_t)in[4 * i + 2] << 16) | ((uint32_t)in[4 * i + 3] << 24);
    return ristretto_decode_affine_w(x, y, w);
}

BPG_HD bool ristretto_decode(ge &p, const uint8_t in[32]) {
    fe x, y;
    if (!ristretto_decode_affine(x, y, in)) return false;
    p.X = x;
    p.Y = y;
    fe_set1(p.Z);
    fe_mul(p.T, x, y);
    return true;
}

// RistrettoPoint::compress, output as 8 LE words
BPG_HD void ristretto_encode_w(uint32_t out[8], const ge &p) {
    fe u1, u2, t, inv, den1, den2, z_inv, ix0, iy0, ench, x, y, den_inv, s, one;
    fe_set1(one);
    fe_add(u1, p.Z, p.Y);
    fe_sub(t, p.Z, p.Y);
    fe_mul(u1, u1, t);
    fe_mul(u2, p.X, p.Y);
    fe_sq(t, u2);
    fe_mul(t, t, u1);
    fe_sqrt_ratio_m1(inv, one, t);
    fe_mul(den1, inv, u1);
    fe_mul(den2, inv, u2);
    fe_mul(z_inv, den1, den2);
    fe_mul(z_inv, z_inv, p.T);
    fe_mul(ix0, p.X, fe_sqrtm1());
    fe_mul(iy0, p.Y, fe_sqrtm1());
    fe_mul(ench, den1, fe_invsqrt_a_md());
    fe_mul(t, p.T, z_inv);
    bool rotate = fe_isnegative(t);
    x = p.X;
    y = p.Y;
    den_inv = den2;
    fe_cmov(x, iy0, rotate);
    fe_cmov(y, ix0, rotate);
    fe_cmov(den_inv, ench, rotate);
    fe_mul(t, x, z_inv);
    if (fe_isnegative(t)) fe_neg(y, y);
    fe_sub(s, p.Z, y);
    fe_mul(s, s, den_inv);
    if (fe_isnegative(s)) fe_neg(s, s);
    fe_towords(out, s);
}
BPG_HD void ristretto_encode(uint8_t out[32], const ge &p) {
    uint32_t w[8];
    ristretto_encode_w(w, p);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)w[i];
        out[4 * i + 1] = (uint8_t)(w[i] >> 8);
        out[4 * i + 2] = (uint8_t)(w[i] >> 16);
        out[4 * i + 3] = (uint8_t)(w[i] >> 24);
    }
}

// elligator_ristretto_flavor (RFC 9496 MAP)
BPG_HD void ristretto_elligator(ge &p, const fe &r0) {
    fe r, u, v, s, sp, c, N, w0, w1, w2, w3, t, one, mone;
    fe_set1(one);
    fe_neg(mone, one);
    fe_sq(r, r0);
    fe_mul(r, r, fe_sqrtm1());
    fe_add(u, r, one);
    fe_mul(u, u, fe_one_m_dsq());
    fe_mul(t, r, fe_d());
    fe_sub(v, mone, t);
    fe_add(t, r, fe_d());
    fe_mul(v, v, t);
    bool was_square = fe_sqrt_ratio_m1(s, u, v);
    fe_mul(sp, s, r0);
    if (!fe_isnegative(sp)) fe_neg(sp, sp);
    c = mone;
    if (!was_square) {
        s = sp;
        c = r;
    }
    fe_sub(t, r, one);
    fe_mul(N, c, t);
    fe_mul(N, N, fe_d_m1_sq());
    fe_sub(N, N, v);
    fe_add(w0, s, s);
    fe_mul(w0, w0, v);
    fe_mul(w1, N, fe_sqrt_ad_m1());
    fe_sq(t, s);
    fe_sub(w2, one, t);
    fe_add(w3, one, t);
    fe_mul(p.X, w0, w3);
    fe_mul(p.Y, w2, w1);
    fe_mul(p.Z, w1, w3);
    fe_mul(p.T, w0, w2);
}

BPG_HD void ristretto_from_uniform_words(ge &p, const uint32_t w[16]) {
    fe r1, r2;
    ge p1, p2;
    fe_fromwords(r1, w);
    fe_fromwords(r2, w + 8);
    ristretto_elligator(p1, r1);
    ristretto_elligator(p2, r2);
    ge_add(p, p1, p2);
}
BPG_HD void ristretto_from_uniform_bytes(ge &p, const uint8_t b[64]) {
    fe r1, r2;
    ge p1, p2;
    fe_frombytes(r1, b);
    fe_frombytes(r2, b + 32);
    ristretto_elligator(p1, r1);
    ristretto_elligator(p2, r2);
    ge_add(p, p1, p2);
}

// RistrettoPoint equality: X1*Y2 == Y1*X2 or X1*X2 == Y1*Y2
BPG_HD bool ristretto_eq(const ge &p, const ge &q) {
    fe a, b, c, d;
    fe_mul(a, p.X, q.Y);
    fe_mul(b, p.Y, q.X);
    fe_mul(c, p.X, q.X);
    fe_mul(d, p.Y, q.Y);
    return fe_eq(a, b) || fe_eq(c, d);
}
BPG_HD bool ristretto_is_identity(const ge &p) {
    // against (0,1,1,0): X*1 == Y*0  or  X*0 == Y*1
    return fe_iszero(p.X) || fe_iszero(p.Y);
}

}  // namespace bpg
