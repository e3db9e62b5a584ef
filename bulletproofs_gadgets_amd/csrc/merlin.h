// merlin.h -- host-side Keccak-f[1600], SHAKE256, SHA3-512, STROBE-128 and the
// Merlin transcript / transcript RNG for the product path.
//
// Replaces the `merlin` crate (Transcript::new at ref: tests/edwards_point.rs:25,59,121,
// 145,200,227,301,330,380,401; src/gadgets/scalar.rs:38,56; sk_know_gadget.rs:48,76) and
// bulletproofs' TranscriptProtocol extension.  Fiat-Shamir is one sequential Keccak
// chain per proof, so it stays on the host: bit-exactness matters here, not throughput.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "sc25519.h"

namespace bpg {

inline uint64_t rotl64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

inline void keccak_f1600(uint64_t s[25]) {
    static const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
        0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
        0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
        0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    // The prover's blinding RNG is one Keccak chain of ~2n permutations (one per drawn
    // scalar), so this is written with the state in locals and every index constant.
    uint64_t a[25];
    for (int i = 0; i < 25; i++) a[i] = s[i];
#define BPG_RHOPI(dst, src, r) b[dst] = (r) ? rotl64(a[src], (r)) : a[src]
    for (int round = 0; round < 24; round++) {
        uint64_t c0 = a[0] ^ a[5] ^ a[10] ^ a[15] ^ a[20], c1 = a[1] ^ a[6] ^ a[11] ^ a[16] ^ a[21];
        uint64_t c2 = a[2] ^ a[7] ^ a[12] ^ a[17] ^ a[22], c3 = a[3] ^ a[8] ^ a[13] ^ a[18] ^ a[23];
        uint64_t c4 = a[4] ^ a[9] ^ a[14] ^ a[19] ^ a[24];
        uint64_t d0 = c4 ^ rotl64(c1, 1), d1 = c0 ^ rotl64(c2, 1), d2 = c1 ^ rotl64(c3, 1), d3 = c2 ^ rotl64(c4, 1),
                 d4 = c3 ^ rotl64(c0, 1);
        for (int y = 0; y < 25; y += 5) {
            a[y] ^= d0;
            a[y + 1] ^= d1;
            a[y + 2] ^= d2;
            a[y + 3] ^= d3;
            a[y + 4] ^= d4;
        }
        uint64_t b[25];
        // rho + pi: B[y + 5((2x+3y) mod 5)] = rot(A[x + 5y], rho[x][y])
        BPG_RHOPI(0, 0, 0);    BPG_RHOPI(10, 1, 1);   BPG_RHOPI(20, 2, 62);  BPG_RHOPI(5, 3, 28);   BPG_RHOPI(15, 4, 27);
        BPG_RHOPI(16, 5, 36);  BPG_RHOPI(1, 6, 44);   BPG_RHOPI(11, 7, 6);   BPG_RHOPI(21, 8, 55);  BPG_RHOPI(6, 9, 20);
        BPG_RHOPI(7, 10, 3);   BPG_RHOPI(17, 11, 10); BPG_RHOPI(2, 12, 43);  BPG_RHOPI(12, 13, 25); BPG_RHOPI(22, 14, 39);
        BPG_RHOPI(23, 15, 41); BPG_RHOPI(8, 16, 45);  BPG_RHOPI(18, 17, 15); BPG_RHOPI(3, 18, 21);  BPG_RHOPI(13, 19, 8);
        BPG_RHOPI(14, 20, 18); BPG_RHOPI(24, 21, 2);  BPG_RHOPI(9, 22, 61);  BPG_RHOPI(19, 23, 56); BPG_RHOPI(4, 24, 14);
        for (int y = 0; y < 25; y += 5) {
            a[y] = b[y] ^ (~b[y + 1] & b[y + 2]);
            a[y + 1] = b[y + 1] ^ (~b[y + 2] & b[y + 3]);
            a[y + 2] = b[y + 2] ^ (~b[y + 3] & b[y + 4]);
            a[y + 3] = b[y + 3] ^ (~b[y + 4] & b[y]);
            a[y + 4] = b[y + 4] ^ (~b[y] & b[y + 1]);
        }
        a[0] ^= RC[round];
    }
#undef BPG_RHOPI
    for (int i = 0; i < 25; i++) s[i] = a[i];
}

// Keccak sponge with byte-granular absorb / squeeze (SHAKE256: rate 136, pad 0x1f; SHA3-512: rate 72, pad 0x06)
class KeccakSponge {
  public:
    KeccakSponge(unsigned rate, uint8_t pad) : rate_(rate), pad_(pad) { memset(st_, 0, sizeof st_); }
    void absorb(const void *data, size_t n) {
        const uint8_t *in = (const uint8_t *)data;
        uint8_t *b = (uint8_t *)st_;
        for (size_t i = 0; i < n; i++) {
            b[pos_++] ^= in[i];
            if (pos_ == rate_) {
                keccak_f1600(st_);
                pos_ = 0;
            }
        }
    }
    void squeeze(void *out, size_t n) {
        uint8_t *o = (uint8_t *)out, *b = (uint8_t *)st_;
        if (!squeezing_) {
            b[pos_] ^= pad_;
            b[rate_ - 1] ^= 0x80;
            keccak_f1600(st_);
            pos_ = 0;
            squeezing_ = true;
        }
        for (size_t i = 0; i < n; i++) {
            if (pos_ == rate_) {
                keccak_f1600(st_);
                pos_ = 0;
            }
            o[i] = b[pos_++];
        }
    }

  private:
    uint64_t st_[25];
    unsigned rate_, pos_ = 0;
    uint8_t pad_;
    bool squeezing_ = false;
};
inline KeccakSponge shake256() { return KeccakSponge(136, 0x1f); }
inline KeccakSponge sha3_512() { return KeccakSponge(72, 0x06); }

// STROBE-128/1600 restricted to the operations Merlin uses (AD, meta-AD, PRF, KEY)
class Strobe128 {
  public:
    explicit Strobe128(const char *protocol_label) {
        memset(st_, 0, sizeof st_);
        const uint8_t hdr[6] = {1, R + 2, 1, 0, 1, 96};
        memcpy(bytes(), hdr, 6);
        memcpy(bytes() + 6, "STROBEv1.0.2", 12);
        keccak_f1600(st_);
        meta_ad(protocol_label, strlen(protocol_label), false);
    }
    void meta_ad(const void *d, size_t n, bool more) {
        begin_op(FLAG_M | FLAG_A, more);
        absorb((const uint8_t *)d, n);
    }
    void ad(const void *d, size_t n, bool more) {
        begin_op(FLAG_A, more);
        absorb((const uint8_t *)d, n);
    }
    void prf(void *d, size_t n, bool more) {
        begin_op(FLAG_I | FLAG_A | FLAG_C, more);
        uint8_t *o = (uint8_t *)d, *b = bytes();
        for (size_t i = 0; i < n; i++) {
            o[i] = b[pos_];
            b[pos_++] = 0;
            if (pos_ == R) run_f();
        }
    }
    void key(const void *d, size_t n, bool more) {
        begin_op(FLAG_A | FLAG_C, more);
        const uint8_t *in = (const uint8_t *)d;
        uint8_t *b = bytes();
        for (size_t i = 0; i < n; i++) {
            b[pos_++] = in[i];
            if (pos_ == R) run_f();
        }
    }

  private:
    static constexpr uint8_t R = 166;
    static constexpr uint8_t FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_M = 16, FLAG_K = 32;
    uint8_t *bytes() { return (uint8_t *)st_; }
    void run_f() {
        uint8_t *b = bytes();
        b[pos_] ^= pos_begin_;
        b[pos_ + 1] ^= 0x04;
        b[R + 1] ^= 0x80;
        keccak_f1600(st_);
        pos_ = 0;
        pos_begin_ = 0;
    }
    void absorb(const uint8_t *d, size_t n) {
        uint8_t *b = bytes();
        for (size_t i = 0; i < n; i++) {
            b[pos_++] ^= d[i];
            if (pos_ == R) run_f();
        }
    }
    void begin_op(uint8_t flags, bool more) {
        if (more) return;
        uint8_t old_begin = pos_begin_;
        pos_begin_ = pos_ + 1;
        cur_flags_ = flags;
        const uint8_t hdr[2] = {old_begin, flags};
        absorb(hdr, 2);
        if ((flags & (FLAG_C | FLAG_K)) && pos_ != 0) run_f();
    }
    uint64_t st_[25];
    uint8_t pos_ = 0, pos_begin_ = 0, cur_flags_ = 0;
};

inline void le32(uint8_t out[4], size_t n) {
    out[0] = (uint8_t)n;
    out[1] = (uint8_t)(n >> 8);
    out[2] = (uint8_t)(n >> 16);
    out[3] = (uint8_t)(n >> 24);
}

// merlin::TranscriptRng
class TranscriptRng {
  public:
    explicit TranscriptRng(const Strobe128 &s) : s_(s) {}
    void rekey_with_witness_bytes(const char *label, const void *w, size_t n) {
        uint8_t len[4];
        le32(len, n);
        s_.meta_ad(label, strlen(label), false);
        s_.meta_ad(len, 4, true);
        s_.key(w, n, false);
    }
    void finalize(const uint8_t external32[32]) {
        s_.meta_ad("rng", 3, false);
        s_.key(external32, 32, false);
    }
    void fill_bytes(void *out, size_t n) {
        uint8_t len[4];
        le32(len, n);
        s_.meta_ad(len, 4, false);
        s_.prf(out, n, false);
    }
    // Scalar::random(&mut rng)
    sc scalar() {
        uint8_t buf[64];
        fill_bytes(buf, 64);
        sc lo, hi, r;
        sc_frombytes(lo, buf);
        sc_frombytes(hi, buf + 32);
        sc_reduce512(r, lo, hi);
        return r;
    }

  private:
    Strobe128 s_;
};

// merlin::Transcript + bulletproofs::transcript::TranscriptProtocol
class Transcript {
  public:
    Transcript(const void *label, size_t n) : s_("Merlin v1.0") { append_message("dom-sep", label, n); }
    void append_message(const char *label, const void *msg, size_t n) {
        uint8_t len[4];
        le32(len, n);
        s_.meta_ad(label, strlen(label), false);
        s_.meta_ad(len, 4, true);
        s_.ad(msg, n, false);
    }
    void append_u64(const char *label, uint64_t x) {
        uint8_t b[8];
        for (int i = 0; i < 8; i++) b[i] = (uint8_t)(x >> (8 * i));
        append_message(label, b, 8);
    }
    void challenge_bytes(const char *label, void *out, size_t n) {
        uint8_t len[4];
        le32(len, n);
        s_.meta_ad(label, strlen(label), false);
        s_.meta_ad(len, 4, true);
        s_.prf(out, n, false);
    }
    sc challenge_scalar(const char *label) {
        uint8_t buf[64];
        challenge_bytes(label, buf, 64);
        sc lo, hi, r;
        sc_frombytes(lo, buf);
        sc_frombytes(hi, buf + 32);
        sc_reduce512(r, lo, hi);
        return r;
    }
    void append_scalar(const char *label, const sc &x) {
        uint8_t b[32];
        sc_tobytes(b, x);
        append_message(label, b, 32);
    }
    void append_point(const char *label, const uint8_t p[32]) { append_message(label, p, 32); }
    // false (and nothing appended) if the point is the identity encoding
    bool validate_and_append_point(const char *label, const uint8_t p[32]) {
        uint8_t acc = 0;
        for (int i = 0; i < 32; i++) acc |= p[i];
        if (!acc) return false;
        append_message(label, p, 32);
        return true;
    }
    void r1cs_domain_sep() { append_message("dom-sep", "r1cs v1", 7); }
    void r1cs_1phase_domain_sep() { append_message("dom-sep", "r1cs-1phase", 11); }
    void innerproduct_domain_sep(uint64_t n) {
        append_message("dom-sep", "ipp v1", 6);
        append_u64("n", n);
    }
    TranscriptRng build_rng() const { return TranscriptRng(s_); }

  private:
    Strobe128 s_;
};

}  // namespace bpg
