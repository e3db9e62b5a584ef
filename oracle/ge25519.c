/* oracle/ge25519.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * edwards25519 in extended coordinates + ristretto255 encode / decode /
 * Elligator, restating curve25519-dalek's EdwardsPoint / RistrettoPoint
 * (== RFC 9496).  Reached from the reference through
 *   Prover::commit -> PedersenGens::commit -> compress   (ref: src/gadgets/point/edwards_point.rs:267)
 *   Verifier::verify -> CompressedRistretto::decompress  (ref: tests/edwards_point.rs:79)
 *   BulletproofGens::new -> from_uniform_bytes           (ref: tests/edwards_point.rs:90)
 */
#include "oracle.h"
#include <string.h>

static const fe FE_D = {{0x34dca135978a3ULL, 0x1a8283b156ebdULL, 0x5e7a26001c029ULL, 0x739c663a03cbbULL, 0x52036cee2b6ffULL}};
static const fe FE_D2 = {{0x69b9426b2f159ULL, 0x35050762add7aULL, 0x3cf44c0038052ULL, 0x6738cc7407977ULL, 0x2406d9dc56dffULL}};
static const fe FE_SQRTM1 = {{0x61b274a0ea0b0ULL, 0xd5a5fc8f189dULL, 0x7ef5e9cbd0c60ULL, 0x78595a6804c9eULL, 0x2b8324804fc1dULL}};
static const fe FE_SQRT_AD_M1 = {{0x7f6a0497b2e1bULL, 0x1836f0a97afd2ULL, 0x7d747f6be7638ULL, 0x456079e7e6498ULL, 0x376931bf2b834ULL}};
static const fe FE_INVSQRT_A_MD = {{0xfdaa805d40eaULL, 0x2eb482e57d339ULL, 0x7610274bc58ULL, 0x6510b613dc8ffULL, 0x786c8905cfaffULL}};
static const fe FE_ONE_M_DSQ = {{0x409c1945fc176ULL, 0x719abc6a1fc4fULL, 0x1c37f90b20684ULL, 0x6bccca55eedfULL, 0x29072a8b2b3eULL}};
static const fe FE_D_M1_SQ = {{0x55aaa44ed4d20ULL, 0x59603c3332635ULL, 0x26d3baf4a7928ULL, 0x120a66e6997a9ULL, 0x5968b37af66c2ULL}};
static const fe FE_BX = {{0x62d608f25d51aULL, 0x412a4b4f6592aULL, 0x75b7171a4b31dULL, 0x1ff60527118feULL, 0x216936d3cd6e5ULL}};
static const fe FE_BY = {{0x6666666666658ULL, 0x4ccccccccccccULL, 0x1999999999999ULL, 0x3333333333333ULL, 0x6666666666666ULL}};
static const fe FE_BT = {{0x68ab3a5b7dda3ULL, 0xeea2a5eadbbULL, 0x2af8df483c27eULL, 0x332b375274732ULL, 0x67875f0fd78b7ULL}};

void ge_identity(ge *p) { fe_0(&p->X); fe_1(&p->Y); fe_1(&p->Z); fe_0(&p->T); }

void ge_basepoint(ge *p) { p->X = FE_BX; p->Y = FE_BY; fe_1(&p->Z); p->T = FE_BT; }

void ge_to_pniels(ge_pniels *r, const ge *p) {
    fe_add(&r->YpX, &p->Y, &p->X);
    fe_sub(&r->YmX, &p->Y, &p->X);
    r->Z = p->Z;
    fe_mul(&r->T2d, &p->T, &FE_D2);
}

/* dalek: &EdwardsPoint + &ProjectiveNielsPoint -> CompletedPoint -> to_extended (4M + 4M) */
void ge_add_pniels(ge *r, const ge *p, const ge_pniels *q) {
    fe YpX, YmX, PP, MM, TT2d, ZZ, ZZ2, X, Y, Z, T;
    fe_add(&YpX, &p->Y, &p->X);
    fe_sub(&YmX, &p->Y, &p->X);
    fe_mul(&PP, &YpX, &q->YpX);
    fe_mul(&MM, &YmX, &q->YmX);
    fe_mul(&TT2d, &p->T, &q->T2d);
    fe_mul(&ZZ, &p->Z, &q->Z);
    fe_add(&ZZ2, &ZZ, &ZZ);
    fe_sub(&X, &PP, &MM);
    fe_add(&Y, &PP, &MM);
    fe_add(&Z, &ZZ2, &TT2d);
    fe_sub(&T, &ZZ2, &TT2d);
    fe_mul(&r->X, &X, &T);
    fe_mul(&r->Y, &Y, &Z);
    fe_mul(&r->Z, &Z, &T);
    fe_mul(&r->T, &X, &Y);
}

void ge_sub_pniels(ge *r, const ge *p, const ge_pniels *q) {
    ge_pniels n;
    n.YpX = q->YmX;
    n.YmX = q->YpX;
    n.Z = q->Z;
    fe_neg(&n.T2d, &q->T2d);
    ge_add_pniels(r, p, &n);
}

void ge_add(ge *r, const ge *p, const ge *q) {
    ge_pniels n;
    ge_to_pniels(&n, q);
    ge_add_pniels(r, p, &n);
}

void ge_neg(ge *r, const ge *p) {
    fe_neg(&r->X, &p->X);
    r->Y = p->Y;
    r->Z = p->Z;
    fe_neg(&r->T, &p->T);
}

void ge_sub(ge *r, const ge *p, const ge *q) {
    ge n;
    ge_neg(&n, q);
    ge_add(r, p, &n);
}

/* dalek: ProjectivePoint::double -> CompletedPoint -> to_extended (4S + 4M) */
void ge_double(ge *r, const ge *p) {
    fe XX, YY, ZZ2, XpY, XpYsq, YYpXX, YYmXX, cX, cT;
    fe_sq(&XX, &p->X);
    fe_sq(&YY, &p->Y);
    fe_sq(&ZZ2, &p->Z);
    fe_add(&ZZ2, &ZZ2, &ZZ2);
    fe_add(&XpY, &p->X, &p->Y);
    fe_sq(&XpYsq, &XpY);
    fe_add(&YYpXX, &YY, &XX);
    fe_sub(&YYmXX, &YY, &XX);
    fe_sub(&cX, &XpYsq, &YYpXX);     /* completed X */
    fe_sub(&cT, &ZZ2, &YYmXX);       /* completed T; completed Y = YYpXX, Z = YYmXX */
    fe_mul(&r->X, &cX, &cT);
    fe_mul(&r->Y, &YYpXX, &YYmXX);
    fe_mul(&r->Z, &YYmXX, &cT);
    fe_mul(&r->T, &cX, &YYpXX);
}

void ge_mul_by_pow_2(ge *r, const ge *p, unsigned k) {
    *r = *p;
    for (unsigned i = 0; i < k; i++) ge_double(r, r);
}

void ge_scalarmult(ge *r, const sc *s, const ge *p) {
    ge acc;
    ge_identity(&acc);
    for (int i = 255; i >= 0; i--) {
        ge_double(&acc, &acc);
        if ((s->v[i >> 6] >> (i & 63)) & 1) ge_add(&acc, &acc, p);
    }
    *r = acc;
}

/* ------------------------------ ristretto255 ------------------------------ */

int ristretto_decode(ge *p, const uint8_t in[32]) {
    fe s, ss, u1, u2, u2sq, v, t, invsqrt, den_x, den_y, x, y, one;
    uint8_t chk[32];
    fe_frombytes(&s, in);
    fe_tobytes(chk, &s);
    if (memcmp(chk, in, 32) != 0) return -1;        /* non-canonical (incl. bit 255 set) */
    if (fe_isnegative(&s)) return -1;
    fe_1(&one);
    fe_sq(&ss, &s);
    fe_sub(&u1, &one, &ss);
    fe_add(&u2, &one, &ss);
    fe_sq(&u2sq, &u2);
    fe_sq(&t, &u1);
    fe_mul(&t, &t, &FE_D);
    fe_neg(&t, &t);
    fe_sub(&v, &t, &u2sq);                          /* -(D*u1^2) - u2^2 */
    fe_mul(&t, &v, &u2sq);
    int ok = fe_sqrt_ratio_m1(&invsqrt, &one, &t);
    fe_mul(&den_x, &invsqrt, &u2);
    fe_mul(&den_y, &invsqrt, &den_x);
    fe_mul(&den_y, &den_y, &v);
    fe_add(&x, &s, &s);
    fe_mul(&x, &x, &den_x);
    if (fe_isnegative(&x)) fe_neg(&x, &x);
    fe_mul(&y, &u1, &den_y);
    fe_mul(&t, &x, &y);
    if (!ok || fe_isnegative(&t) || fe_iszero(&y)) return -1;
    p->X = x; p->Y = y; fe_1(&p->Z); p->T = t;
    return 0;
}

void ristretto_encode(uint8_t out[32], const ge *p) {
    fe u1, u2, t, invsqrt, den1, den2, z_inv, ix0, iy0, ench, x, y, den_inv, s, one;
    fe_1(&one);
    fe_add(&u1, &p->Z, &p->Y);
    fe_sub(&t, &p->Z, &p->Y);
    fe_mul(&u1, &u1, &t);
    fe_mul(&u2, &p->X, &p->Y);
    fe_sq(&t, &u2);
    fe_mul(&t, &t, &u1);
    fe_sqrt_ratio_m1(&invsqrt, &one, &t);
    fe_mul(&den1, &invsqrt, &u1);
    fe_mul(&den2, &invsqrt, &u2);
    fe_mul(&z_inv, &den1, &den2);
    fe_mul(&z_inv, &z_inv, &p->T);
    fe_mul(&ix0, &p->X, &FE_SQRTM1);
    fe_mul(&iy0, &p->Y, &FE_SQRTM1);
    fe_mul(&ench, &den1, &FE_INVSQRT_A_MD);
    fe_mul(&t, &p->T, &z_inv);
    int rotate = fe_isnegative(&t);
    x = p->X; y = p->Y; den_inv = den2;
    fe_cmov(&x, &iy0, rotate);
    fe_cmov(&y, &ix0, rotate);
    fe_cmov(&den_inv, &ench, rotate);
    fe_mul(&t, &x, &z_inv);
    if (fe_isnegative(&t)) fe_neg(&y, &y);
    fe_sub(&s, &p->Z, &y);
    fe_mul(&s, &s, &den_inv);
    if (fe_isnegative(&s)) fe_neg(&s, &s);
    fe_tobytes(out, &s);
}

void ristretto_elligator(ge *p, const fe *r0) {
    fe r, u, v, s, sp, c, N, w0, w1, w2, w3, t, one, mone;
    fe_1(&one);
    fe_neg(&mone, &one);
    fe_sq(&r, r0);
    fe_mul(&r, &r, &FE_SQRTM1);
    fe_add(&u, &r, &one);
    fe_mul(&u, &u, &FE_ONE_M_DSQ);
    fe_mul(&t, &r, &FE_D);
    fe_sub(&v, &mone, &t);
    fe_add(&t, &r, &FE_D);
    fe_mul(&v, &v, &t);
    int was_square = fe_sqrt_ratio_m1(&s, &u, &v);
    fe_mul(&sp, &s, r0);
    if (!fe_isnegative(&sp)) fe_neg(&sp, &sp);      /* -|s*r0| */
    c = mone;
    if (!was_square) { s = sp; c = r; }
    fe_sub(&t, &r, &one);
    fe_mul(&N, &c, &t);
    fe_mul(&N, &N, &FE_D_M1_SQ);
    fe_sub(&N, &N, &v);
    fe_add(&w0, &s, &s);
    fe_mul(&w0, &w0, &v);
    fe_mul(&w1, &N, &FE_SQRT_AD_M1);
    fe_sq(&t, &s);
    fe_sub(&w2, &one, &t);
    fe_add(&w3, &one, &t);
    fe_mul(&p->X, &w0, &w3);
    fe_mul(&p->Y, &w2, &w1);
    fe_mul(&p->Z, &w1, &w3);
    fe_mul(&p->T, &w0, &w2);
}

void ristretto_from_uniform_bytes(ge *p, const uint8_t b[64]) {
    fe r1, r2;
    ge p1, p2;
    fe_frombytes(&r1, b);
    fe_frombytes(&r2, b + 32);
    ristretto_elligator(&p1, &r1);
    ristretto_elligator(&p2, &r2);
    ge_add(p, &p1, &p2);
}

/* dalek RistrettoPoint::ct_eq: X1*Y2 == Y1*X2  or  X1*X2 == Y1*Y2 */
int ristretto_eq(const ge *p, const ge *q) {
    fe a, b, c, d;
    fe_mul(&a, &p->X, &q->Y);
    fe_mul(&b, &p->Y, &q->X);
    fe_mul(&c, &p->X, &q->X);
    fe_mul(&d, &p->Y, &q->Y);
    return fe_eq(&a, &b) | fe_eq(&c, &d);
}

int ristretto_is_identity(const ge *p) {
    ge id;
    ge_identity(&id);
    return ristretto_eq(p, &id);
}

size_t oracle_sizeof_ge(void) { return sizeof(ge); }

void oracle_encode_many(uint8_t *out32, const ge *points, size_t n) {
    for (size_t i = 0; i < n; i++) ristretto_encode(out32 + 32 * i, &points[i]);
}

int oracle_decode_many(ge *points, const uint8_t *in32, size_t n) {
    for (size_t i = 0; i < n; i++)
        if (ristretto_decode(&points[i], in32 + 32 * i) != 0) return -1 - (int)i;
    return 0;
}
