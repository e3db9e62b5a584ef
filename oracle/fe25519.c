/* oracle/fe25519.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * Field arithmetic mod p = 2^255-19 in five 51-bit limbs with 128-bit
 * products: the layout of curve25519-dalek's default serial `u64` backend
 * (FieldElement51), which is what every `Scalar::from_bytes_mod_order(fe.to_bytes())`
 * free curve operation under the reference's prove/verify calls runs on
 * (ref: tests/edwards_point.rs:47,79 -> bulletproofs -> dalek; Cargo.toml:12).
 */
#include "oracle.h"
#include <string.h>

typedef unsigned __int128 u128;
#define MASK51 ((1ULL << 51) - 1)

static inline uint64_t load8(const uint8_t *p) {
    uint64_t r;
    memcpy(&r, p, 8);
    return r;
}

void fe_0(fe *h) { memset(h, 0, sizeof *h); }
void fe_1(fe *h) { memset(h, 0, sizeof *h); h->v[0] = 1; }

void fe_frombytes(fe *h, const uint8_t s[32]) {
    h->v[0] = load8(s) & MASK51;
    h->v[1] = (load8(s + 6) >> 3) & MASK51;
    h->v[2] = (load8(s + 12) >> 6) & MASK51;
    h->v[3] = (load8(s + 19) >> 1) & MASK51;
    h->v[4] = (load8(s + 24) >> 12) & MASK51;   /* drops bit 255 */
}

/* carry once; limbs end below 2^51 + small */
static inline void fe_weak_reduce(fe *h) {
    uint64_t c0 = h->v[0] >> 51, c1 = h->v[1] >> 51, c2 = h->v[2] >> 51, c3 = h->v[3] >> 51, c4 = h->v[4] >> 51;
    h->v[0] &= MASK51; h->v[1] &= MASK51; h->v[2] &= MASK51; h->v[3] &= MASK51; h->v[4] &= MASK51;
    h->v[0] += c4 * 19; h->v[1] += c0; h->v[2] += c1; h->v[3] += c2; h->v[4] += c3;
}

void fe_tobytes(uint8_t s[32], const fe *f) {
    fe t = *f;
    fe_weak_reduce(&t);
    fe_weak_reduce(&t);
    /* now t < 2^255 + tiny; compute q = floor((t + 19) / 2^255) */
    uint64_t q = (t.v[0] + 19) >> 51;
    q = (t.v[1] + q) >> 51;
    q = (t.v[2] + q) >> 51;
    q = (t.v[3] + q) >> 51;
    q = (t.v[4] + q) >> 51;
    t.v[0] += 19 * q;
    uint64_t c;
    c = t.v[0] >> 51; t.v[0] &= MASK51; t.v[1] += c;
    c = t.v[1] >> 51; t.v[1] &= MASK51; t.v[2] += c;
    c = t.v[2] >> 51; t.v[2] &= MASK51; t.v[3] += c;
    c = t.v[3] >> 51; t.v[3] &= MASK51; t.v[4] += c;
    t.v[4] &= MASK51;
    uint64_t w0 = t.v[0] | (t.v[1] << 51);
    uint64_t w1 = (t.v[1] >> 13) | (t.v[2] << 38);
    uint64_t w2 = (t.v[2] >> 26) | (t.v[3] << 25);
    uint64_t w3 = (t.v[3] >> 39) | (t.v[4] << 12);
    memcpy(s, &w0, 8); memcpy(s + 8, &w1, 8); memcpy(s + 16, &w2, 8); memcpy(s + 24, &w3, 8);
}

void fe_add(fe *h, const fe *f, const fe *g) {
    for (int i = 0; i < 5; i++) h->v[i] = f->v[i] + g->v[i];
}

/* f - g + 16p, then one carry pass (inputs may have limbs up to ~2^54) */
void fe_sub(fe *h, const fe *f, const fe *g) {
    fe t;
    t.v[0] = f->v[0] + 36028797018963664ULL - g->v[0];   /* 16*(2^51-19) */
    t.v[1] = f->v[1] + 36028797018963952ULL - g->v[1];   /* 16*(2^51-1)  */
    t.v[2] = f->v[2] + 36028797018963952ULL - g->v[2];
    t.v[3] = f->v[3] + 36028797018963952ULL - g->v[3];
    t.v[4] = f->v[4] + 36028797018963952ULL - g->v[4];
    fe_weak_reduce(&t);
    *h = t;
}

void fe_neg(fe *h, const fe *f) {
    fe z;
    fe_0(&z);
    fe_sub(h, &z, f);
}

void fe_mul(fe *h, const fe *f, const fe *g) {
    const uint64_t *a = f->v, *b = g->v;
    uint64_t b1_19 = b[1] * 19, b2_19 = b[2] * 19, b3_19 = b[3] * 19, b4_19 = b[4] * 19;
    u128 c0 = (u128)a[0] * b[0] + (u128)a[4] * b1_19 + (u128)a[3] * b2_19 + (u128)a[2] * b3_19 + (u128)a[1] * b4_19;
    u128 c1 = (u128)a[1] * b[0] + (u128)a[0] * b[1] + (u128)a[4] * b2_19 + (u128)a[3] * b3_19 + (u128)a[2] * b4_19;
    u128 c2 = (u128)a[2] * b[0] + (u128)a[1] * b[1] + (u128)a[0] * b[2] + (u128)a[4] * b3_19 + (u128)a[3] * b4_19;
    u128 c3 = (u128)a[3] * b[0] + (u128)a[2] * b[1] + (u128)a[1] * b[2] + (u128)a[0] * b[3] + (u128)a[4] * b4_19;
    u128 c4 = (u128)a[4] * b[0] + (u128)a[3] * b[1] + (u128)a[2] * b[2] + (u128)a[1] * b[3] + (u128)a[0] * b[4];
    c1 += (uint64_t)(c0 >> 51); uint64_t o0 = (uint64_t)c0 & MASK51;
    c2 += (uint64_t)(c1 >> 51); uint64_t o1 = (uint64_t)c1 & MASK51;
    c3 += (uint64_t)(c2 >> 51); uint64_t o2 = (uint64_t)c2 & MASK51;
    c4 += (uint64_t)(c3 >> 51); uint64_t o3 = (uint64_t)c3 & MASK51;
    uint64_t carry = (uint64_t)(c4 >> 51); uint64_t o4 = (uint64_t)c4 & MASK51;
    o0 += carry * 19;
    o1 += o0 >> 51; o0 &= MASK51;
    h->v[0] = o0; h->v[1] = o1; h->v[2] = o2; h->v[3] = o3; h->v[4] = o4;
}

void fe_sq(fe *h, const fe *f) { fe_mul(h, f, f); }

static void fe_sqn(fe *h, const fe *f, int n) {
    fe_sq(h, f);
    for (int i = 1; i < n; i++) fe_sq(h, h);
}

/* z^(2^250-1) and z^11, the shared prefix of dalek's invert / pow_p58 */
static void fe_pow22501(fe *t19, fe *t3, const fe *z) {
    fe t0, t1, t2, t4, t5, t6, t7, t8, t9, t10, t11, t12, t13, t14, t15, t16, t17, t18;
    fe_sq(&t0, z);                 /* 2 */
    fe_sqn(&t1, &t0, 2);           /* 8 */
    fe_mul(&t2, z, &t1);           /* 9 */
    fe_mul(t3, &t0, &t2);          /* 11 */
    fe_sq(&t4, t3);                /* 22 */
    fe_mul(&t5, &t2, &t4);         /* 31 = 2^5-1 */
    fe_sqn(&t6, &t5, 5);  fe_mul(&t7, &t6, &t5);     /* 2^10-1 */
    fe_sqn(&t8, &t7, 10); fe_mul(&t9, &t8, &t7);     /* 2^20-1 */
    fe_sqn(&t10, &t9, 20); fe_mul(&t11, &t10, &t9);  /* 2^40-1 */
    fe_sqn(&t12, &t11, 10); fe_mul(&t13, &t12, &t7); /* 2^50-1 */
    fe_sqn(&t14, &t13, 50); fe_mul(&t15, &t14, &t13);/* 2^100-1 */
    fe_sqn(&t16, &t15, 100); fe_mul(&t17, &t16, &t15);/* 2^200-1 */
    fe_sqn(&t18, &t17, 50); fe_mul(t19, &t18, &t13); /* 2^250-1 */
}

void fe_invert(fe *out, const fe *z) {
    fe t19, t3, t20;
    fe_pow22501(&t19, &t3, z);
    fe_sqn(&t20, &t19, 5);         /* 2^255 - 2^5 */
    fe_mul(out, &t20, &t3);        /* 2^255 - 21 */
}

void fe_pow22523(fe *out, const fe *z) {
    fe t19, t3, t20;
    fe_pow22501(&t19, &t3, z);
    fe_sqn(&t20, &t19, 2);         /* 2^252 - 4 */
    fe_mul(out, z, &t20);          /* 2^252 - 3 */
}

int fe_isnegative(const fe *f) {
    uint8_t s[32];
    fe_tobytes(s, f);
    return s[0] & 1;
}

int fe_iszero(const fe *f) {
    uint8_t s[32];
    fe_tobytes(s, f);
    uint8_t r = 0;
    for (int i = 0; i < 32; i++) r |= s[i];
    return r == 0;
}

int fe_eq(const fe *f, const fe *g) {
    uint8_t a[32], b[32];
    fe_tobytes(a, f);
    fe_tobytes(b, g);
    return memcmp(a, b, 32) == 0;
}

void fe_cmov(fe *f, const fe *g, int b) {
    uint64_t m = (uint64_t)0 - (uint64_t)(b != 0);
    for (int i = 0; i < 5; i++) f->v[i] ^= m & (f->v[i] ^ g->v[i]);
}

static const fe FE_SQRTM1 = {{0x61b274a0ea0b0ULL, 0xd5a5fc8f189dULL, 0x7ef5e9cbd0c60ULL, 0x78595a6804c9eULL, 0x2b8324804fc1dULL}};

/* RFC 9496 section 4.2 SQRT_RATIO_M1 == dalek FieldElement::sqrt_ratio_i */
int fe_sqrt_ratio_m1(fe *out, const fe *u, const fe *v) {
    fe v3, v7, r, check, t, neg_u, neg_u_i, r_prime;
    fe_sq(&v3, v); fe_mul(&v3, &v3, v);
    fe_sq(&v7, &v3); fe_mul(&v7, &v7, v);
    fe_mul(&t, u, &v7);
    fe_pow22523(&t, &t);
    fe_mul(&r, u, &v3);
    fe_mul(&r, &r, &t);
    fe_sq(&check, &r); fe_mul(&check, &check, v);
    fe_neg(&neg_u, u);
    fe_mul(&neg_u_i, &neg_u, &FE_SQRTM1);
    int correct = fe_eq(&check, u);
    int flipped = fe_eq(&check, &neg_u);
    int flipped_i = fe_eq(&check, &neg_u_i);
    fe_mul(&r_prime, &r, &FE_SQRTM1);
    fe_cmov(&r, &r_prime, flipped | flipped_i);
    if (fe_isnegative(&r)) fe_neg(&r, &r);
    *out = r;
    return correct | flipped;
}
