// sc25519.h -- scalars mod l = 2^252 + 27742317777372353535851937790883648493
// for the MI355X product path (host + device), eight 32-bit limbs, Montgomery
// reduction with R = 2^256 built from v_mad_u64_u32 rows (l has only five
// non-zero limbs, so a reduction row is five multiply-adds).
//
// Values handed across the API are canonical little-endian 32-byte strings in
// plain (non-Montgomery) form.  Replaces curve25519-dalek's Scalar under
// Prover::prove / Verifier::verify (ref call sites: tests/edwards_point.rs:47,79)
// and Scalar::invert in nonzero_gadget (ref: src/gadgets/scalar.rs:12).
#pragma once
#include "fe25519.h"

namespace bpg {

struct sc {
    uint32_t v[8];
};

#define BPG_SC_CONST(name, ...)                 \
    BPG_HD sc name() {                          \
        sc r = {{__VA_ARGS__}};                 \
        return r;                               \
    }
BPG_SC_CONST(sc_l, 0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0x00000000u, 0x00000000u, 0x00000000u, 0x10000000u)
BPG_SC_CONST(sc_r, 0x8d98951du, 0xd6ec3174u, 0x737dcf70u, 0xc6ef5bf4u, 0xfffffffeu, 0xffffffffu, 0xffffffffu, 0x0fffffffu)
BPG_SC_CONST(sc_rr, 0x449c0f01u, 0xa40611e3u, 0x68859347u, 0xd00e1ba7u, 0x17f5be65u, 0xceec73d2u, 0x7c309a3du, 0x0399411bu)
#define BPG_SC_LFACTOR 0x12547e1bu

BPG_HD sc sc_zero() {
    sc r = {{0, 0, 0, 0, 0, 0, 0, 0}};
    return r;
}
BPG_HD sc sc_one() {
    sc r = {{1, 0, 0, 0, 0, 0, 0, 0}};
    return r;
}
BPG_HD sc sc_from_u64(uint64_t x) {
    sc r = {{(uint32_t)x, (uint32_t)(x >> 32), 0, 0, 0, 0, 0, 0}};
    return r;
}
BPG_HD bool sc_is_zero(const sc &a) {
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) x |= a.v[i];
    return x == 0;
}
BPG_HD bool sc_eq(const sc &a, const sc &b) {
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) x |= a.v[i] ^ b.v[i];
    return x == 0;
}

// t >= l ?
BPG_HD bool sc_geq_l(const uint32_t t[8]) {
    const sc l = sc_l();
    uint32_t b = 0, d;
#pragma unroll
    for (int i = 0; i < 8; i++) d = subb32(t[i], l.v[i], b, b);
    (void)d;
    return b == 0;
}
// t -= l if cond
BPG_HD void sc_csub_l(uint32_t t[8], bool cond) {
    const sc l = sc_l();
    uint32_t b = 0, u[8];
#pragma unroll
    for (int i = 0; i < 8; i++) u[i] = subb32(t[i], l.v[i], b, b);
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = cond ? u[i] : t[i];
}

BPG_HD void sc_add(sc &r, const sc &a, const sc &b) {
    uint32_t t[8], c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = addc32(a.v[i], b.v[i], c, c);
    sc_csub_l(t, sc_geq_l(t));       // a, b < l < 2^253: no carry out
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
}
BPG_HD void sc_sub(sc &r, const sc &a, const sc &b) {
    const sc l = sc_l();
    uint32_t t[8], bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = subb32(a.v[i], b.v[i], bw, bw);
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = addc32(t[i], bw ? l.v[i] : 0u, c, c);
}
BPG_HD void sc_neg(sc &r, const sc &a) {
    sc z = sc_zero();
    sc_sub(r, z, a);
}

// 256x256 -> 512 schoolbook, same row scheme as fe_mul
BPG_HD void mul_wide_8x8(uint32_t acc[16], const uint32_t a[8], const uint32_t b[8]) {
#pragma unroll
    for (int k = 0; k < 16; k++) acc[k] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t lo[8], hi[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint64_t t = (uint64_t)a[i] * b[j] + acc[i + j];
            lo[j] = (uint32_t)t;
            hi[j] = (uint32_t)(t >> 32);
        }
        uint32_t c = 0;
        acc[i] = lo[0];
#pragma unroll
        for (int j = 1; j < 8; j++) acc[i + j] = addc32(lo[j], hi[j - 1], c, c);
        acc[i + 8] = hi[7] + c;
    }
}

// Montgomery reduction: t (512-bit, < l*2^256) -> t / 2^256 mod l, in [0, l)
BPG_HD void sc_montreduce(sc &r, uint32_t t[16]) {
    const sc l = sc_l();
    uint32_t extra = 0;                  // carry beyond t[15]
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t m = t[i] * BPG_SC_LFACTOR;
        // t += m * l << (32 i); l limbs 4,5,6 are zero
        uint64_t p0 = (uint64_t)m * l.v[0] + t[i];
        uint64_t p1 = (uint64_t)m * l.v[1] + t[i + 1];
        uint64_t p2 = (uint64_t)m * l.v[2] + t[i + 2];
        uint64_t p3 = (uint64_t)m * l.v[3] + t[i + 3];
        uint64_t p7 = (uint64_t)m * l.v[7] + t[i + 7];
        uint32_t c = 0;
        t[i] = (uint32_t)p0;             // == 0
        t[i + 1] = addc32((uint32_t)p1, (uint32_t)(p0 >> 32), c, c);
        t[i + 2] = addc32((uint32_t)p2, (uint32_t)(p1 >> 32), c, c);
        t[i + 3] = addc32((uint32_t)p3, (uint32_t)(p2 >> 32), c, c);
        t[i + 4] = addc32(t[i + 4], (uint32_t)(p3 >> 32), c, c);
        t[i + 5] = addc32(t[i + 5], 0, c, c);
        t[i + 6] = addc32(t[i + 6], 0, c, c);
        t[i + 7] = addc32((uint32_t)p7, 0, c, c);
        uint32_t hi7 = (uint32_t)(p7 >> 32);
#pragma unroll
        for (int k = i + 8; k < 16; k++) {
            t[k] = addc32(t[k], (k == i + 8) ? hi7 : 0u, c, c);
        }
        extra += c;
    }
    uint32_t o[8];
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = t[8 + i];
    sc_csub_l(o, extra || sc_geq_l(o));
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = o[i];
}

#if !defined(__HIP_DEVICE_COMPILE__)
// Host form of a*b/R mod l: the same 256-bit values read as four 64-bit limbs, one
// interleaved (CIOS) Montgomery pass with 64x64->128 products.  Circuit capture and the
// transcript's scalar reductions run on the host; the limb layout in memory is unchanged.
inline void sc_montmul_host64(sc &r, const sc &a, const sc &b) {
    typedef unsigned __int128 u128;
    static const uint64_t L[4] = {0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0ULL, 0x1000000000000000ULL};
    const uint64_t LF = 0xd2b51da312547e1bULL;          // -l^-1 mod 2^64
    uint64_t A[4], B[4], t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        A[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
        B[i] = (uint64_t)b.v[2 * i] | ((uint64_t)b.v[2 * i + 1] << 32);
    }
    for (int i = 0; i < 4; i++) {
        u128 acc;
        uint64_t carry = 0;
        for (int j = 0; j < 4; j++) {
            acc = (u128)A[i] * B[j] + t[j] + carry;
            t[j] = (uint64_t)acc;
            carry = (uint64_t)(acc >> 64);
        }
        acc = (u128)t[4] + carry;
        t[4] = (uint64_t)acc;
        t[5] = (uint64_t)(acc >> 64);
        const uint64_t m = t[0] * LF;
        acc = (u128)m * L[0] + t[0];
        carry = (uint64_t)(acc >> 64);
        acc = (u128)m * L[1] + t[1] + carry;
        t[0] = (uint64_t)acc;
        carry = (uint64_t)(acc >> 64);
        acc = (u128)t[2] + carry;                        // L[2] = 0
        t[1] = (uint64_t)acc;
        carry = (uint64_t)(acc >> 64);
        acc = (u128)m * L[3] + t[3] + carry;
        t[2] = (uint64_t)acc;
        carry = (uint64_t)(acc >> 64);
        acc = (u128)t[4] + carry;
        t[3] = (uint64_t)acc;
        t[4] = t[5] + (uint64_t)(acc >> 64);
    }
    // t < 2 l: one conditional subtraction
    uint64_t u[4];
    unsigned __int128 bw = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)t[i] - L[i] - (uint64_t)bw;
        u[i] = (uint64_t)d;
        bw = (d >> 64) & 1;
    }
    const bool ge = t[4] != 0 || bw == 0;
    for (int i = 0; i < 4; i++) {
        const uint64_t w = ge ? u[i] : t[i];
        r.v[2 * i] = (uint32_t)w;
        r.v[2 * i + 1] = (uint32_t)(w >> 32);
    }
}

// Host form of a*b mod l in one pass (plain in, plain out): a 4x4 product of 64-bit limbs, then three folds of the
// part above bit 252 with 2^252 = -c (mod l), c = l - 2^252 < 2^125 (two limbs):
//   P = q 2^252 + lo;  q c = th 2^252 + t_lo;  th c = uh 2^252 + u_lo;  P = lo - t_lo + u_lo - uh c   (mod l)
// 16 + 10 + 6 + 2 = 34 multiplications where two Montgomery passes take 64.  Circuit capture spends half of its time
// here (one product per multiplier), the transcript's challenges and the prover's scalars the rest.
inline void sc_mul_host64(sc &r, const sc &a, const sc &b) {
    typedef unsigned __int128 u128;
    const uint64_t C0 = 0x5812631a5cf5d3edULL, C1 = 0x14def9dea2f79cd6ULL, M60 = (1ULL << 60) - 1;
    uint64_t A[4], B[4], P[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        A[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
        B[i] = (uint64_t)b.v[2 * i] | ((uint64_t)b.v[2 * i + 1] << 32);
    }
    for (int i = 0; i < 4; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < 4; j++) {
            const u128 acc = (u128)A[i] * B[j] + P[i + j] + carry;
            P[i + j] = (uint64_t)acc;
            carry = (uint64_t)(acc >> 64);
        }
        P[i + 4] = carry;
    }
    // x (n limbs) * c -> n + 2 limbs
    auto mul_c = [&](uint64_t *out, const uint64_t *x, int n) {
        uint64_t carry = 0;
        for (int k = 0; k < n; k++) {
            const u128 acc = (u128)x[k] * C0 + carry;
            out[k] = (uint64_t)acc;
            carry = (uint64_t)(acc >> 64);
        }
        out[n] = carry;
        out[n + 1] = 0;
        carry = 0;
        for (int k = 0; k < n; k++) {
            const u128 acc = (u128)x[k] * C1 + out[k + 1] + carry;
            out[k + 1] = (uint64_t)acc;
            carry = (uint64_t)(acc >> 64);
        }
        out[n + 1] += carry;
    };
    uint64_t q[5], t[8], th[3], u[6], w[3];
    for (int k = 0; k < 5; k++) q[k] = (P[3 + k] >> 60) | (P[4 + k] << 4);          // P >> 252 (P[8] = 0)
    mul_c(t, q, 5);                                                                 // < 2^385: 7 limbs
    t[7] = 0;
    for (int k = 0; k < 3; k++) th[k] = (t[3 + k] >> 60) | (t[4 + k] << 4);         // < 2^133
    mul_c(u, th, 3);                                                                // < 2^258: 5 limbs
    const uint64_t uh = (u[3] >> 60) | (u[4] << 4);                                 // < 2^6
    {
        const u128 w0 = (u128)uh * C0, w1 = (u128)uh * C1 + (uint64_t)(w0 >> 64);
        w[0] = (uint64_t)w0;
        w[1] = (uint64_t)w1;
        w[2] = (uint64_t)(w1 >> 64);
    }
    // v = lo + u_lo + 2 l - t_lo - uh c   in (0, 4 l)
    const uint64_t lo[4] = {P[0], P[1], P[2], P[3] & M60}, tl[4] = {t[0], t[1], t[2], t[3] & M60}, ul[4] = {u[0], u[1], u[2], u[3] & M60};
    const uint64_t L2[4] = {C0 << 1, (C1 << 1) | (C0 >> 63), 0, 1ULL << 61}, L1[4] = {C0, C1, 0, 1ULL << 60}, wl[4] = {w[0], w[1], w[2], 0};
    uint64_t v[4];
    __int128 acc = 0;
    for (int k = 0; k < 4; k++) {
        acc += (__int128)lo[k] + ul[k] + L2[k];
        acc -= (__int128)tl[k] + wl[k];
        v[k] = (uint64_t)acc;
        acc >>= 64;
    }
    for (int pass = 0; pass < 2; pass++) {                                          // v >= 2 l ? v -= 2 l;  v >= l ? v -= l
        const uint64_t *m = pass == 0 ? L2 : L1;
        uint64_t d[4];
        __int128 bw = 0;
        for (int k = 0; k < 4; k++) {
            bw += (__int128)v[k] - m[k];
            d[k] = (uint64_t)bw;
            bw >>= 64;
        }
        if (bw == 0)
            for (int k = 0; k < 4; k++) v[k] = d[k];
    }
    for (int i = 0; i < 4; i++) {
        r.v[2 * i] = (uint32_t)v[i];
        r.v[2 * i + 1] = (uint32_t)(v[i] >> 32);
    }
}
#endif

// a*b/R mod l
BPG_HD void sc_montmul(sc &r, const sc &a, const sc &b) {
#if !defined(__HIP_DEVICE_COMPILE__)
    sc_montmul_host64(r, a, b);
#else
    uint32_t t[16];
    mul_wide_8x8(t, a.v, b.v);
    sc_montreduce(r, t);
#endif
}
BPG_HD void sc_to_mont(sc &r, const sc &a) { sc_montmul(r, a, sc_rr()); }
BPG_HD void sc_from_mont(sc &r, const sc &a) { sc_montmul(r, a, sc_one()); }

// plain * plain -> plain
BPG_HD void sc_mul(sc &r, const sc &a, const sc &b) {
#if !defined(__HIP_DEVICE_COMPILE__)
    sc_mul_host64(r, a, b);
#else
    sc t;
    sc_montmul(t, a, b);
    sc_montmul(r, t, sc_rr());
#endif
}
// r = a*b + c
BPG_HD void sc_muladd(sc &r, const sc &a, const sc &b, const sc &c) {
    sc t;
    sc_mul(t, a, b);
    sc_add(r, t, c);
}

// any 256-bit string -> mod l (Scalar::from_bytes_mod_order)
BPG_HD void sc_reduce256(sc &r, const sc &a) { sc_montmul(r, a, sc_r()); }
// 512-bit little-endian -> mod l (Scalar::from_bytes_mod_order_wide)
BPG_HD void sc_reduce512(sc &r, const sc &lo, const sc &hi) {
    sc a, b;
    sc_montmul(a, lo, sc_r());
    sc_montmul(b, hi, sc_rr());
    sc_add(r, a, b);
}

BPG_HD void sc_frombytes(sc &r, const uint8_t s[32]) {
#pragma unroll
    for (int i = 0; i < 8; i++)
        r.v[i] = (uint32_t)s[4 * i] | ((uint32_t)s[4 * i + 1] << 8) | ((uint32_t)s[4 * i + 2] << 16) | ((uint32_t)s[4 * i + 3] << 24);
}
BPG_HD void sc_tobytes(uint8_t s[32], const sc &a) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        s[4 * i] = (uint8_t)a.v[i];
        s[4 * i + 1] = (uint8_t)(a.v[i] >> 8);
        s[4 * i + 2] = (uint8_t)(a.v[i] >> 16);
        s[4 * i + 3] = (uint8_t)(a.v[i] >> 24);
    }
}
BPG_HD bool sc_is_canonical(const sc &a) { return !sc_geq_l(a.v); }

// a^(l-2) (Scalar::invert); returns 0 for 0 like dalek
BPG_HD void sc_invert(sc &r, const sc &a) {
    const uint32_t e[8] = {0x5cf5d3ebu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0, 0, 0, 0x10000000u};
    sc x, acc = sc_r();
    sc_to_mont(x, a);
    for (int i = 252; i >= 0; i--) {
        sc_montmul(acc, acc, acc);
        if ((e[i >> 5] >> (i & 31)) & 1) sc_montmul(acc, acc, x);
    }
    sc_from_mont(r, acc);
}

}  // namespace bpg
