// fe25519.h -- arithmetic mod p = 2^255 - 19 for the MI355X product path.
//
// Representation: eight 32-bit limbs, little-endian, value in [0, 2^256)
// ("weakly reduced": congruent mod p, not necessarily below p).  Every
// partial product is one v_mad_u64_u32 (32x32 + 64 -> 64) and every carry
// chain is v_add_co_u32 / v_addc_co_u32; 2^256 = 38 (mod p) folds the high
// half back.  The same source compiles for the host (used for the short
// sequential tails: window Horner, compress, decompress).
//
// Replaces, for the hot path, curve25519-dalek's FieldElement (the serial
// backend under every curve operation of Prover::prove / Verifier::verify,
// ref call sites: tests/edwards_point.rs:47,79; Cargo.toml:12).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BPG_HD __host__ __device__ __forceinline__

namespace bpg {

struct fe {
    uint32_t v[8];
};

BPG_HD uint32_t addc32(uint32_t a, uint32_t b, uint32_t cin, uint32_t &cout) {
    return __builtin_addc(a, b, cin, &cout);
}
BPG_HD uint32_t subb32(uint32_t a, uint32_t b, uint32_t bin, uint32_t &bout) {
    return __builtin_subc(a, b, bin, &bout);
}

BPG_HD void fe_set0(fe &r) {
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
}
BPG_HD void fe_set1(fe &r) {
    fe_set0(r);
    r.v[0] = 1;
}
BPG_HD void fe_setu32(fe &r, uint32_t x) {
    fe_set0(r);
    r.v[0] = x;
}

// r = a + b (mod p), weakly reduced
BPG_HD void fe_add(fe &r, const fe &a, const fe &b) {
    uint32_t c = 0, t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = addc32(a.v[i], b.v[i], c, c);
    // fold the 2^256 carry: += 38*c, twice (second fold can only touch limb 0)
    uint32_t c2 = 0;
    t[0] = addc32(t[0], c ? 38u : 0u, 0, c2);
#pragma unroll
    for (int i = 1; i < 8; i++) t[i] = addc32(t[i], 0, c2, c2);
    t[0] += c2 ? 38u : 0u;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
}

// r = a - b (mod p), weakly reduced
BPG_HD void fe_sub(fe &r, const fe &a, const fe &b) {
    uint32_t c = 0, t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = subb32(a.v[i], b.v[i], c, c);
    uint32_t c2 = 0;
    t[0] = subb32(t[0], c ? 38u : 0u, 0, c2);
#pragma unroll
    for (int i = 1; i < 8; i++) t[i] = subb32(t[i], 0, c2, c2);
    t[0] -= c2 ? 38u : 0u;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
}

BPG_HD void fe_neg(fe &r, const fe &a) {
    fe z;
    fe_set0(z);
    fe_sub(r, z, a);
}

// acc[0..15] (512-bit) -> r (weakly reduced), using 2^256 = 38
BPG_HD void fe_reduce512(fe &r, const uint32_t acc[16]) {
    uint32_t lo[8], hi[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        uint64_t t = (uint64_t)acc[8 + j] * 38u + acc[j];
        lo[j] = (uint32_t)t;
        hi[j] = (uint32_t)(t >> 32);
    }
    uint32_t w[8], c = 0;
    w[0] = lo[0];
#pragma unroll
    for (int j = 1; j < 8; j++) w[j] = addc32(lo[j], hi[j - 1], c, c);
    uint32_t top = hi[7] + c;                 // <= 38
    uint32_t c2 = 0;
    w[0] = addc32(w[0], top * 38u, 0, c2);
#pragma unroll
    for (int j = 1; j < 8; j++) w[j] = addc32(w[j], 0, c2, c2);
    w[0] += c2 ? 38u : 0u;
#pragma unroll
    for (int j = 0; j < 8; j++) r.v[j] = w[j];
}

// r = a * b (mod p).  64 v_mad_u64_u32 for the product (row-wise; the eight
// products of a row are independent, one carry chain per row), 9 for the fold.
BPG_HD void fe_mul(fe &r, const fe &a, const fe &b) {
    uint32_t acc[16];
    {
        uint32_t lo[8], hi[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint64_t t = (uint64_t)a.v[0] * b.v[j];
            lo[j] = (uint32_t)t;
            hi[j] = (uint32_t)(t >> 32);
        }
        uint32_t c = 0;
        acc[0] = lo[0];
#pragma unroll
        for (int j = 1; j < 8; j++) acc[j] = addc32(lo[j], hi[j - 1], c, c);
        acc[8] = hi[7] + c;
    }
#pragma unroll
    for (int i = 1; i < 8; i++) {
        uint32_t lo[8], hi[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint64_t t = (uint64_t)a.v[i] * b.v[j] + acc[i + j];
            lo[j] = (uint32_t)t;
            hi[j] = (uint32_t)(t >> 32);
        }
        uint32_t c = 0;
        acc[i] = lo[0];
#pragma unroll
        for (int j = 1; j < 8; j++) acc[i + j] = addc32(lo[j], hi[j - 1], c, c);
        acc[i + 8] = hi[7] + c;
    }
    fe_reduce512(r, acc);
}

// r = a^2 (mod p): 36 products; off-diagonal sum doubled by a shift.
BPG_HD void fe_sq(fe &r, const fe &a) {
    // off-diagonal part: sum_{i<j} a_i a_j 2^(32(i+j)), rows i = 0..6
    uint32_t od[16];
#pragma unroll
    for (int k = 0; k < 16; k++) od[k] = 0;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        uint32_t lo[8], hi[8];
#pragma unroll
        for (int j = i + 1; j < 8; j++) {
            uint64_t t = (uint64_t)a.v[i] * a.v[j] + od[i + j];
            lo[j] = (uint32_t)t;
            hi[j] = (uint32_t)(t >> 32);
        }
        uint32_t c = 0;
        od[2 * i + 1] = lo[i + 1];
#pragma unroll
        for (int j = i + 2; j < 8; j++) od[i + j] = addc32(lo[j], hi[j - 1], c, c);
        od[i + 8] = hi[7] + c;
    }
    // acc = 2*od + diag
    uint32_t acc[16];
    uint32_t prev = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        acc[k] = (od[k] << 1) | (prev >> 31);
        prev = od[k];
    }
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t t = (uint64_t)a.v[i] * a.v[i];
        acc[2 * i] = addc32(acc[2 * i], (uint32_t)t, c, c);
        acc[2 * i + 1] = addc32(acc[2 * i + 1], (uint32_t)(t >> 32), c, c);
    }
    fe_reduce512(r, acc);
}

BPG_HD void fe_sqn(fe &r, const fe &a, int n) {
    fe_sq(r, a);
    for (int i = 1; i < n; i++) fe_sq(r, r);
}

// r = a * small (mod p)
BPG_HD void fe_mul_small(fe &r, const fe &a, uint32_t s) {
    uint32_t lo[8], hi[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        uint64_t t = (uint64_t)a.v[j] * s;
        lo[j] = (uint32_t)t;
        hi[j] = (uint32_t)(t >> 32);
    }
    uint32_t w[8], c = 0;
    w[0] = lo[0];
#pragma unroll
    for (int j = 1; j < 8; j++) w[j] = addc32(lo[j], hi[j - 1], c, c);
    uint64_t top = (uint64_t)(hi[7] + c) * 38u;   // hi[7]+c < 2^32, top < 2^38
    uint32_t c2 = 0;
    w[0] = addc32(w[0], (uint32_t)top, 0, c2);
    w[1] = addc32(w[1], (uint32_t)(top >> 32), c2, c2);
#pragma unroll
    for (int j = 2; j < 8; j++) w[j] = addc32(w[j], 0, c2, c2);
    w[0] += c2 ? 38u : 0u;
#pragma unroll
    for (int j = 0; j < 8; j++) r.v[j] = w[j];
}

// canonical little-endian bytes (value fully reduced below p)
BPG_HD void fe_canon(fe &r, const fe &a) {
    // fold bit 255: x = lo255 + 19*b255, twice is enough to get below 2^255 + small
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = a.v[i];
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {
        uint32_t top = t[7] >> 31;
        t[7] &= 0x7fffffffu;
        uint32_t c = 0;
        t[0] = addc32(t[0], top * 19u, 0, c);
#pragma unroll
        for (int i = 1; i < 8; i++) t[i] = addc32(t[i], 0, c, c);
    }
    // now t < 2^255 + 19; subtract p if t >= p: compute t + 19, look at bit 255
    uint32_t u[8], c = 0;
    u[0] = addc32(t[0], 19u, 0, c);
#pragma unroll
    for (int i = 1; i < 8; i++) u[i] = addc32(t[i], 0, c, c);
    uint32_t ge = u[7] >> 31;     // t + 19 >= 2^255  <=>  t >= p
    u[7] &= 0x7fffffffu;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = ge ? u[i] : t[i];
}

BPG_HD void fe_tobytes(uint8_t s[32], const fe &a) {
    fe c;
    fe_canon(c, a);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        s[4 * i] = (uint8_t)c.v[i];
        s[4 * i + 1] = (uint8_t)(c.v[i] >> 8);
        s[4 * i + 2] = (uint8_t)(c.v[i] >> 16);
        s[4 * i + 3] = (uint8_t)(c.v[i] >> 24);
    }
}

// word (little-endian u32 x 8) forms of to/from bytes: no byte arrays on the device
BPG_HD void fe_towords(uint32_t w[8], const fe &a) {
    fe c;
    fe_canon(c, a);
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = c.v[i];
}
BPG_HD void fe_fromwords(fe &r, const uint32_t w[8]) {
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = w[i];
    r.v[7] &= 0x7fffffffu;
}

// ignores bit 255 (dalek FieldElement::from_bytes)
BPG_HD void fe_frombytes(fe &r, const uint8_t s[32]) {
#pragma unroll
    for (int i = 0; i < 8; i++)
        r.v[i] = (uint32_t)s[4 * i] | ((uint32_t)s[4 * i + 1] << 8) | ((uint32_t)s[4 * i + 2] << 16) | ((uint32_t)s[4 * i + 3] << 24);
    r.v[7] &= 0x7fffffffu;
}

BPG_HD bool fe_iszero(const fe &a) {
    fe c;
    fe_canon(c, a);
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) x |= c.v[i];
    return x == 0;
}
BPG_HD bool fe_isnegative(const fe &a) {
    fe c;
    fe_canon(c, a);
    return c.v[0] & 1;
}
BPG_HD bool fe_eq(const fe &a, const fe &b) {
    fe d;
    fe_sub(d, a, b);
    return fe_iszero(d);
}
BPG_HD void fe_cmov(fe &r, const fe &a, bool b) {
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = b ? a.v[i] : r.v[i];
}

// z^(2^250-1) and z^11
BPG_HD void fe_pow22501(fe &t19, fe &t3, const fe &z) {
    fe t0, t1, t2, t4, t5, t6, t7, t9, t11, t13, t15, t17;
    fe_sq(t0, z);
    fe_sqn(t1, t0, 2);
    fe_mul(t2, z, t1);
    fe_mul(t3, t0, t2);
    fe_sq(t4, t3);
    fe_mul(t5, t2, t4);
    fe_sqn(t6, t5, 5);    fe_mul(t7, t6, t5);
    fe_sqn(t6, t7, 10);   fe_mul(t9, t6, t7);
    fe_sqn(t6, t9, 20);   fe_mul(t11, t6, t9);
    fe_sqn(t6, t11, 10);  fe_mul(t13, t6, t7);
    fe_sqn(t6, t13, 50);  fe_mul(t15, t6, t13);
    fe_sqn(t6, t15, 100); fe_mul(t17, t6, t15);
    fe_sqn(t6, t17, 50);  fe_mul(t19, t6, t13);
}
BPG_HD void fe_invert(fe &r, const fe &z) {
    fe t19, t3, t20;
    fe_pow22501(t19, t3, z);
    fe_sqn(t20, t19, 5);
    fe_mul(r, t20, t3);
}
BPG_HD void fe_pow22523(fe &r, const fe &z) {
    fe t19, t3, t20;
    fe_pow22501(t19, t3, z);
    fe_sqn(t20, t19, 2);
    fe_mul(r, z, t20);
}

}  // namespace bpg
