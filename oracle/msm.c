/* oracle/msm.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * The multiscalar-multiplication schedules of curve25519-dalek 2.x
 * (backend/serial/scalar_mul/{straus,pippenger}.rs), restated:
 *   msm_straus_ct    -- Straus::multiscalar_mul: radix-16 signed digits, 8-entry
 *                       ProjectiveNiels lookup table per point, 64 x (4 doublings + n adds).
 *                       This is what Prover::prove uses for A_I, A_O, S
 *                       (ref call sites: tests/edwards_point.rs:47,134,215,320,391).
 *   msm_straus_vt    -- Straus::optional_multiscalar_mul: width-5 NAF, odd multiples.
 *   msm_pippenger_vt -- Pippenger::optional_multiscalar_mul: w = 6 (<500), 7 (<800), 8.
 *   msm_vartime      -- RistrettoPoint::vartime_multiscalar_mul dispatch (n < 190 ? Straus : Pippenger);
 *                       used by InnerProductProof::create (L, R, G/H folds) and
 *                       Verifier::verify (ref: tests/edwards_point.rs:79).
 * The table selections here are not constant-time (this is a checker and a
 * timing baseline, not a product); the group results are identical.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

static void radix16(int8_t out[64], const uint8_t s[32]) {
    for (int i = 0; i < 32; i++) {
        out[2 * i] = s[i] & 15;
        out[2 * i + 1] = (s[i] >> 4) & 15;
    }
    for (int i = 0; i < 63; i++) {
        int8_t carry = (int8_t)((out[i] + 8) >> 4);
        out[i] -= (int8_t)(carry << 4);
        out[i + 1] += carry;
    }
}

static void naf(int8_t out[256], const uint8_t s[32], unsigned w) {
    uint64_t x[5] = {0, 0, 0, 0, 0};
    memcpy(x, s, 32);
    memset(out, 0, 256);
    uint64_t width = 1ULL << w, mask = width - 1;
    unsigned pos = 0;
    uint64_t carry = 0;
    while (pos < 256) {
        unsigned idx = pos / 64, bit = pos % 64;
        uint64_t buf;
        if (bit < 64 - w) buf = x[idx] >> bit;
        else buf = (x[idx] >> bit) | (x[idx + 1] << (64 - bit));
        uint64_t window = carry + (buf & mask);
        if ((window & 1) == 0) { pos += 1; continue; }
        if (window < width / 2) { carry = 0; out[pos] = (int8_t)window; }
        else { carry = 1; out[pos] = (int8_t)((int64_t)window - (int64_t)width); }
        pos += w;
    }
}

static unsigned radix2w_count(unsigned w) { return (256 + w - 1) / w + (w == 8 ? 1 : 0); }

static void radix2w(int8_t *digits, const uint8_t s[32], unsigned w) {
    uint64_t x[4];
    memcpy(x, s, 32);
    uint64_t radix = 1ULL << w, mask = radix - 1, carry = 0;
    unsigned count = (256 + w - 1) / w;
    memset(digits, 0, 43);
    for (unsigned i = 0; i < count; i++) {
        unsigned off = i * w, idx = off / 64, bit = off % 64;
        uint64_t buf;
        if (bit < 64 - w || idx == 3) buf = x[idx] >> bit;
        else buf = (x[idx] >> bit) | (x[idx + 1] << (64 - bit));
        uint64_t coef = carry + (buf & mask);
        carry = (coef + radix / 2) >> w;
        digits[i] = (int8_t)((int64_t)coef - (int64_t)(carry << w));
    }
    if (w == 8) digits[count] += (int8_t)carry;
    else digits[count - 1] += (int8_t)(carry << w);
}

void msm_naive(ge *r, const uint8_t *scalars32, const ge *points, size_t n) {
    ge acc, t;
    ge_identity(&acc);
    for (size_t i = 0; i < n; i++) {
        sc s;
        memcpy(s.v, scalars32 + 32 * i, 32);
        ge_scalarmult(&t, &s, &points[i]);
        ge_add(&acc, &acc, &t);
    }
    *r = acc;
}

void msm_straus_ct(ge *r, const uint8_t *scalars32, const ge *points, size_t n) {
    ge_pniels *tables = malloc(n * 8 * sizeof(ge_pniels));   /* [P, 2P, ..., 8P] */
    int8_t *digits = malloc(n * 64);
    for (size_t i = 0; i < n; i++) {
        ge m = points[i];
        ge_to_pniels(&tables[8 * i], &m);
        for (int j = 1; j < 8; j++) {
            ge_add_pniels(&m, &points[i], &tables[8 * i + j - 1]);
            ge_to_pniels(&tables[8 * i + j], &m);
        }
        radix16(digits + 64 * i, scalars32 + 32 * i);
    }
    ge q;
    ge_identity(&q);
    for (int j = 63; j >= 0; j--) {
        ge_mul_by_pow_2(&q, &q, 4);
        for (size_t i = 0; i < n; i++) {
            int d = digits[64 * i + j];
            /* LookupTable::select: |d|-th entry, conditionally negated; d = 0 -> identity (still one add in dalek) */
            if (d > 0) ge_add_pniels(&q, &q, &tables[8 * i + d - 1]);
            else if (d < 0) ge_sub_pniels(&q, &q, &tables[8 * i - d - 1]);
            else {
                ge_pniels id;
                fe_1(&id.YpX); fe_1(&id.YmX); fe_1(&id.Z); fe_0(&id.T2d);
                ge_add_pniels(&q, &q, &id);
            }
        }
    }
    free(tables);
    free(digits);
    *r = q;
}

void msm_straus_vt(ge *r, const uint8_t *scalars32, const ge *points, size_t n) {
    ge_pniels *tables = malloc((n ? n : 1) * 8 * sizeof(ge_pniels));  /* [P, 3P, ..., 15P] */
    int8_t *nafs = malloc((n ? n : 1) * 256);
    for (size_t i = 0; i < n; i++) {
        ge p2, m = points[i];
        ge_double(&p2, &points[i]);
        ge_to_pniels(&tables[8 * i], &m);
        for (int j = 1; j < 8; j++) {
            ge_add_pniels(&m, &p2, &tables[8 * i + j - 1]);
            ge_to_pniels(&tables[8 * i + j], &m);
        }
        naf(nafs + 256 * i, scalars32 + 32 * i, 5);
    }
    ge q;
    ge_identity(&q);
    for (int j = 255; j >= 0; j--) {
        ge_double(&q, &q);
        for (size_t i = 0; i < n; i++) {
            int d = nafs[256 * i + j];
            if (d > 0) ge_add_pniels(&q, &q, &tables[8 * i + d / 2]);
            else if (d < 0) ge_sub_pniels(&q, &q, &tables[8 * i + (-d) / 2]);
        }
    }
    free(tables);
    free(nafs);
    *r = q;
}

void msm_pippenger_vt(ge *r, const uint8_t *scalars32, const ge *points, size_t n) {
    unsigned w = n < 500 ? 6 : (n < 800 ? 7 : 8);
    unsigned max_digit = 1u << w, count = radix2w_count(w), nbuckets = max_digit / 2;
    int8_t *digits = malloc((n ? n : 1) * 43);
    ge_pniels *pn = malloc((n ? n : 1) * sizeof(ge_pniels));
    ge *buckets = malloc(nbuckets * sizeof(ge));
    for (size_t i = 0; i < n; i++) {
        radix2w(digits + 43 * i, scalars32 + 32 * i, w);
        ge_to_pniels(&pn[i], &points[i]);
    }
    ge total;
    int first = 1;
    for (int col = (int)count - 1; col >= 0; col--) {
        for (unsigned b = 0; b < nbuckets; b++) ge_identity(&buckets[b]);
        for (size_t i = 0; i < n; i++) {
            int d = digits[43 * i + col];
            if (d > 0) ge_add_pniels(&buckets[d - 1], &buckets[d - 1], &pn[i]);
            else if (d < 0) ge_sub_pniels(&buckets[-d - 1], &buckets[-d - 1], &pn[i]);
        }
        ge inter = buckets[nbuckets - 1], sum = buckets[nbuckets - 1];
        for (int b = (int)nbuckets - 2; b >= 0; b--) {
            ge_add(&inter, &inter, &buckets[b]);
            ge_add(&sum, &sum, &inter);
        }
        if (first) { total = sum; first = 0; }
        else {
            ge_mul_by_pow_2(&total, &total, w);
            ge_add(&total, &total, &sum);
        }
    }
    free(digits); free(pn); free(buckets);
    *r = total;
}

void msm_vartime(ge *r, const uint8_t *scalars32, const ge *points, size_t n) {
    if (n < 190) msm_straus_vt(r, scalars32, points, n);
    else msm_pippenger_vt(r, scalars32, points, n);
}

static void dispatch(ge *r, const uint8_t *scalars32, const ge *points, size_t n, int schedule) {
    switch (schedule) {
    case 0: msm_vartime(r, scalars32, points, n); break;
    case 1: msm_straus_ct(r, scalars32, points, n); break;
    case 2: msm_straus_vt(r, scalars32, points, n); break;
    case 3: msm_pippenger_vt(r, scalars32, points, n); break;
    default: msm_naive(r, scalars32, points, n); break;
    }
}

void oracle_msm_ge(uint8_t out32[32], const uint8_t *scalars32, const ge *points, size_t n, int schedule) {
    ge r;
    dispatch(&r, scalars32, points, n, schedule);
    ristretto_encode(out32, &r);
}

/* un-normalised extended result X|Y|Z|T (4 x 32 B canonical field encodings): the
 * partial-sum wire format of point-range sharding (include/bpg.h, bpg_msm_wait_partial) */
void oracle_msm_ge_ext(uint8_t out128[128], const uint8_t *scalars32, const ge *points, size_t n, int schedule) {
    ge r;
    dispatch(&r, scalars32, points, n, schedule);
    fe_tobytes(out128, &r.X);
    fe_tobytes(out128 + 32, &r.Y);
    fe_tobytes(out128 + 64, &r.Z);
    fe_tobytes(out128 + 96, &r.T);
}

int oracle_msm_compressed(uint8_t out32[32], const uint8_t *scalars32, const uint8_t *points32, size_t n, int schedule) {
    ge *pts = malloc((n ? n : 1) * sizeof(ge));
    int rc = oracle_decode_many(pts, points32, n);
    if (rc == 0) oracle_msm_ge(out32, scalars32, pts, n, schedule);
    free(pts);
    return rc;
}

void oracle_fold_points(ge *out, const ge *P, size_t half, const uint8_t *s_lo32, const uint8_t *s_hi32, size_t stride) {
    for (size_t i = 0; i < half; i++) {
        uint8_t s[64];
        ge pts[2] = {P[i], P[half + i]};
        memcpy(s, s_lo32 + 32 * i * stride, 32);
        memcpy(s + 32, s_hi32 + 32 * i * stride, 32);
        ge r;
        msm_vartime(&r, s, pts, 2);
        out[i] = r;
    }
}
