/* oracle/keccak_merlin.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * Keccak-f[1600], SHAKE256, SHA3-512; STROBE-128 as used by merlin 2.0
 * (strobe.rs) and merlin's Transcript / TranscriptRng (transcript.rs), restated
 * from the published specs.  Reached from the reference at every
 * `Transcript::new(b"...")` (ref: tests/edwards_point.rs:25,121,200,301;
 * src/gadgets/scalar.rs:38; src/gadgets/sk_knowledge/sk_know_gadget.rs:48) and
 * inside Prover/Verifier.  Also the bulletproofs generator chain
 * (generators.rs: SHAKE256("GeneratorsChain" || label)) and
 * PedersenGens::default() (ref: tests/edwards_point.rs:89-90).
 * Little-endian host assumed.
 */
#include "oracle.h"
#include <string.h>

static const uint64_t RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int ROTC[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
static const int PILN[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};

static inline uint64_t rotl64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

void keccak_f1600(uint64_t st[25]) {
    uint64_t bc[5], t;
    for (int round = 0; round < 24; round++) {
        for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        for (int i = 0; i < 5; i++) {
            t = bc[(i + 4) % 5] ^ rotl64(bc[(i + 1) % 5], 1);
            for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
        }
        t = st[1];
        for (int i = 0; i < 24; i++) {
            int j = PILN[i];
            bc[0] = st[j];
            st[j] = rotl64(t, ROTC[i]);
            t = bc[0];
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = st[j + i];
            for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        st[0] ^= RC[round];
    }
}

void shake256_init(keccak_sponge *k) { memset(k, 0, sizeof *k); k->rate = 136; }
void sha3_512_init(keccak_sponge *k) { memset(k, 0, sizeof *k); k->rate = 72; }

void keccak_absorb(keccak_sponge *k, const uint8_t *in, size_t n) {
    uint8_t *b = (uint8_t *)k->st;
    for (size_t i = 0; i < n; i++) {
        b[k->pos++] ^= in[i];
        if (k->pos == k->rate) { keccak_f1600(k->st); k->pos = 0; }
    }
}

static void keccak_pad(keccak_sponge *k, uint8_t dsep) {
    uint8_t *b = (uint8_t *)k->st;
    b[k->pos] ^= dsep;
    b[k->rate - 1] ^= 0x80;
    keccak_f1600(k->st);
    k->pos = 0;
    k->squeezing = 1;
}

void shake256_squeeze(keccak_sponge *k, uint8_t *out, size_t n) {
    uint8_t *b = (uint8_t *)k->st;
    if (!k->squeezing) keccak_pad(k, 0x1f);
    for (size_t i = 0; i < n; i++) {
        if (k->pos == k->rate) { keccak_f1600(k->st); k->pos = 0; }
        out[i] = b[k->pos++];
    }
}

void sha3_512_final(keccak_sponge *k, uint8_t out[64]) {
    keccak_pad(k, 0x06);
    memcpy(out, k->st, 64);
}

/* ------------------------------- STROBE-128 ------------------------------- */
#define STROBE_R 166
#define FLAG_I 1
#define FLAG_A 2
#define FLAG_C 4
#define FLAG_T 8
#define FLAG_M 16
#define FLAG_K 32

static void strobe_run_f(strobe128 *s) {
    s->st[s->pos] ^= s->pos_begin;
    s->st[s->pos + 1] ^= 0x04;
    s->st[STROBE_R + 1] ^= 0x80;
    uint64_t w[25];
    memcpy(w, s->st, 200);
    keccak_f1600(w);
    memcpy(s->st, w, 200);
    s->pos = 0;
    s->pos_begin = 0;
}

static void strobe_absorb(strobe128 *s, const uint8_t *d, size_t n) {
    for (size_t i = 0; i < n; i++) {
        s->st[s->pos++] ^= d[i];
        if (s->pos == STROBE_R) strobe_run_f(s);
    }
}

static void strobe_overwrite(strobe128 *s, const uint8_t *d, size_t n) {
    for (size_t i = 0; i < n; i++) {
        s->st[s->pos++] = d[i];
        if (s->pos == STROBE_R) strobe_run_f(s);
    }
}

static void strobe_squeeze(strobe128 *s, uint8_t *d, size_t n) {
    for (size_t i = 0; i < n; i++) {
        d[i] = s->st[s->pos];
        s->st[s->pos++] = 0;
        if (s->pos == STROBE_R) strobe_run_f(s);
    }
}

static void strobe_begin_op(strobe128 *s, uint8_t flags, int more) {
    if (more) return;                 /* continuing: merlin asserts cur_flags == flags */
    uint8_t old_begin = s->pos_begin;
    s->pos_begin = s->pos + 1;
    s->cur_flags = flags;
    uint8_t hdr[2] = {old_begin, flags};
    strobe_absorb(s, hdr, 2);
    if ((flags & (FLAG_C | FLAG_K)) && s->pos != 0) strobe_run_f(s);
}

static void strobe_meta_ad(strobe128 *s, const uint8_t *d, size_t n, int more) { strobe_begin_op(s, FLAG_M | FLAG_A, more); strobe_absorb(s, d, n); }
static void strobe_ad(strobe128 *s, const uint8_t *d, size_t n, int more) { strobe_begin_op(s, FLAG_A, more); strobe_absorb(s, d, n); }
static void strobe_prf(strobe128 *s, uint8_t *d, size_t n, int more) { strobe_begin_op(s, FLAG_I | FLAG_A | FLAG_C, more); strobe_squeeze(s, d, n); }
static void strobe_key(strobe128 *s, const uint8_t *d, size_t n, int more) { strobe_begin_op(s, FLAG_A | FLAG_C, more); strobe_overwrite(s, d, n); }

static void strobe_new(strobe128 *s, const uint8_t *label, size_t n) {
    memset(s, 0, sizeof *s);
    static const uint8_t hdr[6] = {1, STROBE_R + 2, 1, 0, 1, 96};
    memcpy(s->st, hdr, 6);
    memcpy(s->st + 6, "STROBEv1.0.2", 12);
    uint64_t w[25];
    memcpy(w, s->st, 200);
    keccak_f1600(w);
    memcpy(s->st, w, 200);
    strobe_meta_ad(s, label, n, 0);
}

static void le32(uint8_t out[4], size_t n) {
    out[0] = (uint8_t)n; out[1] = (uint8_t)(n >> 8); out[2] = (uint8_t)(n >> 16); out[3] = (uint8_t)(n >> 24);
}

void merlin_init(merlin_transcript *t, const uint8_t *label, size_t n) {
    strobe_new(&t->s, (const uint8_t *)"Merlin v1.0", 11);
    merlin_append_message(t, "dom-sep", label, n);
}

void merlin_append_message(merlin_transcript *t, const char *label, const uint8_t *msg, size_t n) {
    uint8_t len[4];
    le32(len, n);
    strobe_meta_ad(&t->s, (const uint8_t *)label, strlen(label), 0);
    strobe_meta_ad(&t->s, len, 4, 1);
    strobe_ad(&t->s, msg, n, 0);
}

void merlin_append_u64(merlin_transcript *t, const char *label, uint64_t x) {
    uint8_t b[8];
    for (int i = 0; i < 8; i++) b[i] = (uint8_t)(x >> (8 * i));
    merlin_append_message(t, label, b, 8);
}

void merlin_challenge_bytes(merlin_transcript *t, const char *label, uint8_t *out, size_t n) {
    uint8_t len[4];
    le32(len, n);
    strobe_meta_ad(&t->s, (const uint8_t *)label, strlen(label), 0);
    strobe_meta_ad(&t->s, len, 4, 1);
    strobe_prf(&t->s, out, n, 0);
}

void merlin_challenge_scalar(merlin_transcript *t, const char *label, uint8_t out32[32]) {
    uint8_t buf[64];
    sc s;
    merlin_challenge_bytes(t, label, buf, 64);
    sc_frombytes_wide(&s, buf);
    sc_tobytes(out32, &s);
}

void merlin_rng_build(merlin_rng *r, const merlin_transcript *t) { r->s = t->s; }

void merlin_rng_rekey(merlin_rng *r, const char *label, const uint8_t *w, size_t n) {
    uint8_t len[4];
    le32(len, n);
    strobe_meta_ad(&r->s, (const uint8_t *)label, strlen(label), 0);
    strobe_meta_ad(&r->s, len, 4, 1);
    strobe_key(&r->s, w, n, 0);
}

void merlin_rng_finalize(merlin_rng *r, const uint8_t rng_bytes32[32]) {
    strobe_meta_ad(&r->s, (const uint8_t *)"rng", 3, 0);
    strobe_key(&r->s, rng_bytes32, 32, 0);
}

void merlin_rng_fill(merlin_rng *r, uint8_t *out, size_t n) {
    uint8_t len[4];
    le32(len, n);
    strobe_meta_ad(&r->s, len, 4, 0);
    strobe_prf(&r->s, out, n, 0);
}

void merlin_rng_scalar(merlin_rng *r, uint8_t out32[32]) {
    uint8_t buf[64];
    sc s;
    merlin_rng_fill(r, buf, 64);
    sc_frombytes_wide(&s, buf);
    sc_tobytes(out32, &s);
}

void merlin_rng_scalars(merlin_rng *r, uint8_t *out32, size_t count) {
    for (size_t i = 0; i < count; i++) merlin_rng_scalar(r, out32 + 32 * i);
}

/* --------------------------- bulletproofs gens ---------------------------- */

void pedersen_gens_default(ge *B, ge *B_blinding) {
    uint8_t enc[32], h[64];
    keccak_sponge k;
    ge_basepoint(B);
    ristretto_encode(enc, B);
    sha3_512_init(&k);
    keccak_absorb(&k, enc, 32);
    sha3_512_final(&k, h);
    ristretto_from_uniform_bytes(B_blinding, h);
}

static void gens_chain(ge *out, size_t n, char tag) {
    uint8_t label[5] = {(uint8_t)tag, 0, 0, 0, 0};   /* tag || u32le(party index = 0) */
    keccak_sponge k;
    shake256_init(&k);
    keccak_absorb(&k, (const uint8_t *)"GeneratorsChain", 15);
    keccak_absorb(&k, label, 5);
    for (size_t i = 0; i < n; i++) {
        uint8_t u[64];
        shake256_squeeze(&k, u, 64);
        ristretto_from_uniform_bytes(&out[i], u);
    }
}

void bulletproof_gens(ge *G, ge *H, size_t n) {
    gens_chain(G, n, 'G');
    gens_chain(H, n, 'H');
}
