/* oracle/sc25519.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * Scalars mod l = 2^252 + 27742317777372353535851937790883648493
 * (curve25519-dalek `Scalar`; reached from the reference at e.g.
 * ref: src/gadgets/scalar.rs:12 `q.invert()`, ref: src/gadgets/point/edwards_point.rs:21
 * `Scalar::from_bytes_mod_order`, and all of prove/verify).
 * Four 64-bit limbs, Montgomery multiplication with R = 2^256.  Values at the
 * API are canonical (in [0,l), plain representation).
 */
#include "oracle.h"
#include <string.h>

typedef unsigned __int128 u128;

static const uint64_t SC_L[4]  = {0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0x0ULL, 0x1000000000000000ULL};
static const uint64_t SC_LFACTOR = 0xd2b51da312547e1bULL;  /* -l^{-1} mod 2^64 */
static const uint64_t SC_R[4]  = {0xd6ec31748d98951dULL, 0xc6ef5bf4737dcf70ULL, 0xfffffffffffffffeULL, 0x0fffffffffffffffULL};
static const uint64_t SC_RR[4] = {0xa40611e3449c0f01ULL, 0xd00e1ba768859347ULL, 0xceec73d217f5be65ULL, 0x0399411b7c309a3dULL};

static int geq_l(const uint64_t a[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > SC_L[i]) return 1;
        if (a[i] < SC_L[i]) return 0;
    }
    return 1;
}

static void sub_l(uint64_t a[4]) {
    u128 borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 t = (u128)a[i] - SC_L[i] - borrow;
        a[i] = (uint64_t)t;
        borrow = (t >> 64) & 1;
    }
}

/* a*b*R^-1 mod l; requires a*b < l*R; output in [0,l) */
static void montmul(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a[i] * b[j] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * SC_LFACTOR;
        c = (u128)m * SC_L[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * SC_L[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    uint64_t out[4] = {t[0], t[1], t[2], t[3]};
    if (t[4] || geq_l(out)) sub_l(out);
    memcpy(r, out, 32);
}

void sc_frombytes_mod_order(sc *r, const uint8_t s[32]) {
    uint64_t a[4];
    memcpy(a, s, 32);
    montmul(r->v, a, SC_R);     /* a * R * R^-1 = a mod l, a < 2^256 <= R */
}

void sc_frombytes_wide(sc *r, const uint8_t s[64]) {
    uint64_t lo[4], hi[4], a[4], b[4];
    memcpy(lo, s, 32);
    memcpy(hi, s + 32, 32);
    montmul(a, lo, SC_R);       /* lo mod l */
    montmul(b, hi, SC_RR);      /* hi * R mod l */
    sc x, y;
    memcpy(x.v, a, 32); memcpy(y.v, b, 32);
    sc_add(r, &x, &y);
}

int sc_is_canonical_bytes(const uint8_t s[32]) {
    uint64_t a[4];
    memcpy(a, s, 32);
    return !geq_l(a);
}

void sc_tobytes(uint8_t s[32], const sc *a) { memcpy(s, a->v, 32); }

void sc_add(sc *r, const sc *a, const sc *b) {
    uint64_t t[4];
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)a->v[i] + b->v[i];
        t[i] = (uint64_t)c;
        c >>= 64;
    }
    if (geq_l(t)) sub_l(t);     /* a,b < l < 2^253 so no carry out */
    memcpy(r->v, t, 32);
}

void sc_neg(sc *r, const sc *a) {
    uint64_t t[4];
    if ((a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0) { memset(r->v, 0, 32); return; }
    u128 borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)SC_L[i] - a->v[i] - borrow;
        t[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
    memcpy(r->v, t, 32);
}

void sc_sub(sc *r, const sc *a, const sc *b) {
    sc nb;
    sc_neg(&nb, b);
    sc_add(r, a, &nb);
}

void sc_mul(sc *r, const sc *a, const sc *b) {
    uint64_t t[4];
    montmul(t, a->v, b->v);     /* ab/R */
    montmul(r->v, t, SC_RR);    /* ab   */
}

void sc_invert(sc *r, const sc *a) {
    /* a^(l-2), square-and-multiply in Montgomery form */
    static const uint64_t LM2[4] = {0x5812631a5cf5d3ebULL, 0x14def9dea2f79cd6ULL, 0x0ULL, 0x1000000000000000ULL};
    uint64_t x[4], acc[4];
    montmul(x, a->v, SC_RR);          /* aR */
    memcpy(acc, SC_R, 32);            /* 1R */
    for (int i = 252; i >= 0; i--) {
        montmul(acc, acc, acc);
        if ((LM2[i >> 6] >> (i & 63)) & 1) montmul(acc, acc, x);
    }
    static const uint64_t ONE[4] = {1, 0, 0, 0};
    montmul(r->v, acc, ONE);
}
