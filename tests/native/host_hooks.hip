// tests/native/host_hooks.hip -- TEST-ONLY shared library.
// Exposes the HOST instantiation of the product's host+device arithmetic
// headers (fe25519.h / sc25519.h / ge25519.h) so that the CPU test-suite can
// check the exact source the GPU kernels are built from against Python big
// integers and the oracle, without a GPU.  No HIP runtime call is made here.
#include <cstring>
#include "../../bulletproofs_gadgets_amd/csrc/ge25519.h"
#include "../../bulletproofs_gadgets_amd/csrc/sc25519.h"
#include "../../bulletproofs_gadgets_amd/csrc/merlin.h"

using namespace bpg;

static void fe_in(fe &r, const uint8_t *b) {
    for (int i = 0; i < 8; i++)
        r.v[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
}
static void sc_in(sc &r, const uint8_t *b) { sc_frombytes(r, b); }

extern "C" {

// raw 256-bit in (weakly reduced inputs allowed), canonical 32 bytes out
void hh_fe_mul(uint8_t *out, const uint8_t *a, const uint8_t *b) { fe x, y, r; fe_in(x, a); fe_in(y, b); fe_mul(r, x, y); fe_tobytes(out, r); }
void hh_fe_sq(uint8_t *out, const uint8_t *a) { fe x, r; fe_in(x, a); fe_sq(r, x); fe_tobytes(out, r); }
void hh_fe_add(uint8_t *out, const uint8_t *a, const uint8_t *b) { fe x, y, r; fe_in(x, a); fe_in(y, b); fe_add(r, x, y); fe_tobytes(out, r); }
void hh_fe_sub(uint8_t *out, const uint8_t *a, const uint8_t *b) { fe x, y, r; fe_in(x, a); fe_in(y, b); fe_sub(r, x, y); fe_tobytes(out, r); }
void hh_fe_mul_small(uint8_t *out, const uint8_t *a, uint32_t s) { fe x, r; fe_in(x, a); fe_mul_small(r, x, s); fe_tobytes(out, r); }
void hh_fe_invert(uint8_t *out, const uint8_t *a) { fe x, r; fe_in(x, a); fe_invert(r, x); fe_tobytes(out, r); }
// raw (un-canonicalised) limbs of a product, to test the weak-reduction invariant
void hh_fe_mul_raw(uint8_t *out, const uint8_t *a, const uint8_t *b) { fe x, y, r; fe_in(x, a); fe_in(y, b); fe_mul(r, x, y); memcpy(out, r.v, 32); }

void hh_sc_mul(uint8_t *out, const uint8_t *a, const uint8_t *b) { sc x, y, r; sc_in(x, a); sc_in(y, b); sc_mul(r, x, y); sc_tobytes(out, r); }
void hh_sc_add(uint8_t *out, const uint8_t *a, const uint8_t *b) { sc x, y, r; sc_in(x, a); sc_in(y, b); sc_add(r, x, y); sc_tobytes(out, r); }
void hh_sc_sub(uint8_t *out, const uint8_t *a, const uint8_t *b) { sc x, y, r; sc_in(x, a); sc_in(y, b); sc_sub(r, x, y); sc_tobytes(out, r); }
void hh_sc_invert(uint8_t *out, const uint8_t *a) { sc x, r; sc_in(x, a); sc_invert(r, x); sc_tobytes(out, r); }
void hh_sc_reduce256(uint8_t *out, const uint8_t *a) { sc x, r; sc_in(x, a); sc_reduce256(r, x); sc_tobytes(out, r); }
void hh_sc_reduce512(uint8_t *out, const uint8_t *a) { sc lo, hi, r; sc_in(lo, a); sc_in(hi, a + 32); sc_reduce512(r, lo, hi); sc_tobytes(out, r); }

int hh_ristretto_decode_encode(uint8_t *out, const uint8_t *in) {
    ge p;
    if (!ristretto_decode(p, in)) return -1;
    ristretto_encode(out, p);
    return 0;
}
void hh_from_uniform(uint8_t *out, const uint8_t *in64) { ge p; ristretto_from_uniform_bytes(p, in64); ristretto_encode(out, p); }
// out = compress(2*P + Q) through double / add / madd paths
int hh_group_ops(uint8_t *out_dbl_add, uint8_t *out_madd, uint8_t *out_msub, const uint8_t *P32, const uint8_t *Q32) {
    ge p, q, r;
    fe qx, qy;
    if (!ristretto_decode(p, P32) || !ristretto_decode_affine(qx, qy, Q32)) return -1;
    q.X = qx; q.Y = qy; fe_set1(q.Z); fe_mul(q.T, qx, qy);
    ge_double(r, p);
    ge_add(r, r, q);
    ristretto_encode(out_dbl_add, r);
    ge_aniels n;
    ge_affine_to_aniels(n, qx, qy);
    ge_madd(r, p, n, false);
    ristretto_encode(out_madd, r);
    ge_madd(r, p, n, true);
    ristretto_encode(out_msub, r);
    return 0;
}

// SHAKE256(in) -> out (host Keccak of the product: generator chain, transcript)
void hh_shake256(uint8_t *out, size_t out_len, const uint8_t *in, size_t in_len) {
    KeccakSponge x = shake256();
    x.absorb(in, in_len);
    x.squeeze(out, out_len);
}
// both compiled bodies of Keccak-f[1600] (baseline x86-64 and BMI) on the same state; 0 = equal.
// which: 0 = generic only -> out, 1 = dispatching entry point -> out
void hh_keccak_f(uint64_t *state, int which) {
    if (which == 0) keccak_f1600_generic(state);
    else keccak_f1600(state);
}
// TranscriptRng: Transcript::new(label).build_rng().rekey("v_blinding", witness32).finalize(ext32) -> n scalars
void hh_transcript_rng(uint8_t *out, size_t n, const uint8_t *label, size_t label_len, const uint8_t *witness32, const uint8_t *ext32) {
    Transcript tr(label, label_len);
    TranscriptRng rng = tr.build_rng();
    rng.rekey_with_witness_bytes("v_blinding", witness32, 32);
    rng.finalize(ext32);
    for (size_t i = 0; i < n; i++) {
        sc s = rng.scalar();
        sc_tobytes(out + 32 * i, s);
    }
}
// eight interleaved Keccak states st[25][8]; generic != 0 forces the per-lane scalar fallback
void hh_keccak_x8(uint64_t *st, int generic) {
    uint64_t (*s)[8] = (uint64_t (*)[8])st;
    if (generic) {
        for (int l = 0; l < 8; l++) {
            uint64_t t[25];
            for (int i = 0; i < 25; i++) t[i] = s[i][l];
            keccak_f1600_generic(t);
            for (int i = 0; i < 25; i++) s[i][l] = t[i];
        }
    } else {
        keccak_f1600_x8(s);
    }
}
// eight TranscriptRngs (witness / external randomness k of 8) drawn in lock-step: n scalars per lane through
// TranscriptRngLanes8, then, after store(), `tail` more from each lane's own TranscriptRng.
// out: lane-major, (n + tail) * 32 bytes per lane.  Returns 0 if the lanes could not be loaded.
int hh_transcript_rng_lanes(uint8_t *out, size_t n, size_t tail, const uint8_t *label, size_t label_len, const uint8_t *witness32x8,
                            const uint8_t *ext32x8) {
    std::vector<TranscriptRng> rngs;
    Transcript tr(label, label_len);
    for (int l = 0; l < 8; l++) {
        rngs.push_back(tr.build_rng());
        rngs[l].rekey_with_witness_bytes("v_blinding", witness32x8 + 32 * l, 32);
        rngs[l].finalize(ext32x8 + 32 * l);
    }
    TranscriptRng *ptr[8];
    for (int l = 0; l < 8; l++) ptr[l] = &rngs[l];
    TranscriptRngLanes8 lanes;
    if (!lanes.load(ptr)) return 0;
    for (size_t i = 0; i < n; i++) {
        sc s[8];
        lanes.scalars(s);
        for (int l = 0; l < 8; l++) sc_tobytes(out + ((n + tail) * l + i) * 32, s[l]);
    }
    lanes.store(ptr, 8);
    for (int l = 0; l < 8; l++)
        for (size_t i = 0; i < tail; i++) {
            sc s = rngs[l].scalar();
            sc_tobytes(out + ((n + tail) * l + n + i) * 32, s);
        }
    return 1;
}
}
