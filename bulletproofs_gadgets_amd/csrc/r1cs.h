// r1cs.h -- host side of the product's R1CS layer: Variable / LinearCombination /
// ConstraintSystem (the interface the reference's gadgets are written against:
// `cs: &mut dyn ConstraintSystem`, ref: src/gadgets/point/edwards_point.rs:28),
// Prover, Verifier, generators, the proof object, and zerocaf's Sonny curve for
// witness values.  Circuit capture is host work (it is the gadget API that "stays
// as it is"); prove() / verify() drive the HIP kernels (r1cs.hip, msm.hip).
//
// Mirrors bulletproofs::r1cs (git develop, feature yoloproofs; ref: Cargo.toml:15-18):
//   Prover::new/commit/prove      ref: tests/edwards_point.rs:28,31,47
//   Verifier::new/commit/verify   ref: tests/edwards_point.rs:61,63,79
//   ConstraintSystem::multiply / allocate_multiplier / constrain
//                                 ref: edwards_point.rs:45; scalar.rs:10-21; boolean.rs:20
#pragma once
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <stdexcept>
#include <unordered_map>
#include <atomic>
#include <thread>
#include <vector>

#include "bpg_internal.h"
#include "merlin.h"

namespace bpg {

// ---- scalar convenience (host) ------------------------------------------------
struct Scalar {
    sc v;
    Scalar() : v(sc_zero()) {}
    Scalar(const sc &x) : v(x) {}
    Scalar(uint64_t x) : v(sc_from_u64(x)) {}
    static Scalar from_bytes(const uint8_t b[32]) {      // Scalar::from_bytes_mod_order
        sc t, r;
        sc_frombytes(t, b);
        sc_reduce256(r, t);
        return Scalar(r);
    }
    void to_bytes(uint8_t b[32]) const { sc_tobytes(b, v); }
    Scalar operator+(const Scalar &o) const { sc r; sc_add(r, v, o.v); return r; }
    Scalar operator-(const Scalar &o) const { sc r; sc_sub(r, v, o.v); return r; }
    Scalar operator*(const Scalar &o) const { sc r; sc_mul(r, v, o.v); return r; }
    Scalar operator-() const { sc r; sc_neg(r, v); return r; }
    Scalar &operator+=(const Scalar &o) { sc_add(v, v, o.v); return *this; }
    Scalar &operator-=(const Scalar &o) { sc_sub(v, v, o.v); return *this; }
    Scalar &operator*=(const Scalar &o) { sc_mul(v, v, o.v); return *this; }
    bool operator==(const Scalar &o) const { return sc_eq(v, o.v); }
    bool is_zero() const { return sc_is_zero(v); }
    Scalar invert() const { sc r; sc_invert(r, v); return r; }
};

// ---- variables and linear combinations ---------------------------------------
enum VarKind : uint32_t { VAR_COMMITTED = 0, VAR_LEFT = 1, VAR_RIGHT = 2, VAR_OUTPUT = 3, VAR_ONE = 4 };
struct Variable {
    uint32_t kind, index;
    static Variable One() { return Variable{VAR_ONE, 0}; }
};

struct Term {
    Variable var;
    Scalar coeff;
};

// A linear combination as the gadgets build them: tens of thousands of short-lived objects with one to six terms each,
// nearly all with coefficient +1 or -1 (ref: src/gadgets/point/edwards_point.rs:55-79: sums and differences of
// wires).  A term is therefore 8 bytes -- the variable packed as kind << 28 | index and a small signed coefficient --
// and only a coefficient that is not a small integer (a public coordinate, the curve constant d) takes a 32-byte
// slot in a side table, in order of appearance.  Negating or copying a term of +-1 is integer work; capture
// evaluates and compares such terms without touching a scalar.  Up to INL terms and INL_BIG large coefficients live
// inside the object; beyond that everything moves to the heap.
class LinearCombination {
  public:
    struct Entry {
        uint32_t var;       // kind << 28 | index
        int32_t k;          // the coefficient if it is a small integer (|k| < 2^30), else BIG: the next slot of the side table
    };
    static constexpr int32_t BIG = INT32_MIN;
    static constexpr uint32_t INL = 7, INL_BIG = 4;

    LinearCombination() {}
    LinearCombination(const Variable &v) {
        e_[0] = Entry{v.kind << 28 | v.index, 1};
        n_ = 1;
    }
    LinearCombination(const Scalar &s) { push(Variable::One(), s); }
    LinearCombination(const LinearCombination &o) { copy_from(o); }
    LinearCombination(LinearCombination &&o) noexcept {
        copy_inline(o);
        h_ = o.h_;
        o.h_ = nullptr;
        o.n_ = o.nbig_ = 0;
    }
    LinearCombination &operator=(const LinearCombination &o) {
        if (this != &o) {
            release();
            copy_from(o);
        }
        return *this;
    }
    LinearCombination &operator=(LinearCombination &&o) noexcept {
        if (this != &o) {
            release();
            copy_inline(o);
            h_ = o.h_;
            o.h_ = nullptr;
            o.n_ = o.nbig_ = 0;
        }
        return *this;
    }
    ~LinearCombination() { release(); }

    size_t size() const { return n_; }
    size_t num_big() const { return nbig_; }
    const Entry *entries() const { return h_ ? h_->e.data() : e_; }
    const sc *bigs() const { return h_ ? h_->big.data() : big_; }

    // small integer value of a scalar (|k| < 2^30: k or l - |k|), else BIG
    static int32_t small_of(const sc &c) {
        if ((c.v[1] | c.v[2] | c.v[3] | c.v[4] | c.v[5] | c.v[6] | c.v[7]) == 0) return c.v[0] < (1u << 30) ? (int32_t)c.v[0] : BIG;
        // l - x, 0 < x < 2^30: only the lowest limb differs from l's (l's lowest limb is 0x5cf5d3ed > 2^30: no borrow)
        if (c.v[7] == 0x10000000u && (c.v[4] | c.v[5] | c.v[6]) == 0 && c.v[3] == 0x14def9deu && c.v[2] == 0xa2f79cd6u &&
            c.v[1] == 0x5812631au && c.v[0] < 0x5cf5d3edu && 0x5cf5d3edu - c.v[0] < (1u << 30))
            return -(int32_t)(0x5cf5d3edu - c.v[0]);
        return BIG;
    }
    static sc scalar_of(int32_t k) {
        sc r = sc_from_u64((uint64_t)(k < 0 ? -(int64_t)k : (int64_t)k));
        if (k < 0) sc_neg(r, r);
        return r;
    }
    void push(const Variable &v, const Scalar &c) {
        const int32_t k = small_of(c.v);
        if (k != BIG) push_small(v.kind << 28 | v.index, k);
        else push_big(v.kind << 28 | v.index, c.v);
    }
    void reserve(size_t terms) {
        if (terms > INL) to_heap(terms);
    }
    // the i-th term, materialized (tests, export; capture walks entries() / bigs() instead)
    Term term(size_t i) const {
        const Entry *e = entries();
        size_t b = 0;
        for (size_t j = 0; j < i; j++) b += e[j].k == BIG;
        const Variable v{e[i].var >> 28, e[i].var & 0x0fffffffu};
        return Term{v, e[i].k == BIG ? Scalar(bigs()[b]) : Scalar(scalar_of(e[i].k))};
    }

    LinearCombination &operator+=(const LinearCombination &o) {
        append(o, false);
        return *this;
    }
    LinearCombination &operator-=(const LinearCombination &o) {
        append(o, true);
        return *this;
    }
    LinearCombination operator+(const LinearCombination &o) const { LinearCombination r = *this; r.append(o, false); return r; }
    LinearCombination operator-(const LinearCombination &o) const { LinearCombination r = *this; r.append(o, true); return r; }
    LinearCombination operator-() const { LinearCombination r; r.append(*this, true); return r; }

  private:
    struct Heap {
        std::vector<Entry> e;
        std::vector<sc> big;
    };
    void copy_inline(const LinearCombination &o) {
        n_ = o.n_;
        nbig_ = o.nbig_;
        if (!o.h_) {
            memcpy(e_, o.e_, o.n_ * sizeof(Entry));
            memcpy(static_cast<void *>(big_), o.big_, o.nbig_ * sizeof(sc));
        }
    }
    void copy_from(const LinearCombination &o) {
        copy_inline(o);
        h_ = o.h_ ? new Heap(*o.h_) : nullptr;
    }
    void release() {
        delete h_;
        h_ = nullptr;
        n_ = nbig_ = 0;
    }
    void to_heap(size_t want_terms) {
        if (!h_) {
            h_ = new Heap();
            h_->e.assign(e_, e_ + n_);
            h_->big.assign(big_, big_ + nbig_);
        }
        h_->e.reserve(want_terms);
    }
    void push_small(uint32_t var, int32_t k) {
        if (h_) {
            h_->e.push_back(Entry{var, k});
        } else if (n_ < INL) {
            e_[n_] = Entry{var, k};
        } else {
            to_heap(2 * INL);
            h_->e.push_back(Entry{var, k});
        }
        n_++;
    }
    void push_big(uint32_t var, const sc &c) {
        if (!h_ && (n_ >= INL || nbig_ >= INL_BIG)) to_heap(2 * INL);
        if (h_) {
            h_->e.push_back(Entry{var, BIG});
            h_->big.push_back(c);
        } else {
            e_[n_] = Entry{var, BIG};
            big_[nbig_] = c;
        }
        n_++;
        nbig_++;
    }
    // *this += o or *this -= o (o may be *this)
    void append(const LinearCombination &o, bool negate) {
        const uint32_t m = o.n_, mb = o.nbig_;
        if (!h_ && !o.h_ && n_ + m <= INL && nbig_ + mb <= INL_BIG) {          // the common case: all inline
            for (uint32_t i = 0; i < m; i++) {
                const Entry t = o.e_[i];
                e_[n_ + i] = Entry{t.var, negate && t.k != BIG ? -t.k : t.k};
            }
            for (uint32_t i = 0; i < mb; i++) {
                if (negate) sc_neg(big_[nbig_ + i], o.big_[i]);
                else big_[nbig_ + i] = o.big_[i];
            }
            n_ += m;
            nbig_ += mb;
            return;
        }
        to_heap((size_t)n_ + m);
        std::vector<Entry> oe(o.entries(), o.entries() + m);               // (copies: o may be *this, and the vectors may move)
        std::vector<sc> ob(o.bigs(), o.bigs() + mb);
        for (const Entry &t : oe) h_->e.push_back(Entry{t.var, negate && t.k != BIG ? -t.k : t.k});
        for (const sc &c : ob) {
            sc x = c;
            if (negate) sc_neg(x, c);
            h_->big.push_back(x);
        }
        n_ += m;
        nbig_ += mb;
    }

    uint32_t n_ = 0, nbig_ = 0;
    Entry e_[INL];
    sc big_[INL_BIG];
    Heap *h_ = nullptr;
};
using LC = LinearCombination;

struct R1CSError : std::runtime_error {
    int code;
    R1CSError(int c, const char *what) : std::runtime_error(what), code(c) {}
};

// ---- constraint system --------------------------------------------------------
// Constraints are stored flat (structure of arrays) so prove()/verify() can hand
// them to the GPU flattening kernel without re-walking LC objects:
//   term t: variable (kind, index), coefficient id (dictionary: circuits use a
//   handful of distinct coefficients), constraint q = number of constrain() calls before it.
// The structure of a captured circuit: which variable, which coefficient id and which constraint every
// linear-combination term belongs to, in capture order.  Immutable once published; shared between the
// circuit-structure cache (r1cs.hip: FlatCached) and the constraint systems that follow it.
struct CircuitKey {
    std::vector<uint32_t> term_var, term_coeff, term_q;
    uint32_t num_constraints = 0, num_coeffs = 0;
    uint32_t n_mult = 0;            // multipliers of the circuit (a hint: a follower reserves its witness vectors)
};

struct ConstraintStore {
    std::vector<uint32_t> term_var;      // kind << 28 | index   (index < 2^28)
    std::vector<uint32_t> term_coeff;    // index into coeffs
    std::vector<uint32_t> term_q;        // constraint number
    std::vector<sc> coeffs;
    std::unordered_map<uint64_t, std::vector<uint32_t>> coeff_lookup;   // hash -> ids
    uint32_t num_constraints = 0;
    uint32_t id_one = 0xffffffffu, id_minus_one = 0xffffffffu;       // almost every coefficient is +1 or -1
    std::vector<int32_t> small_of_id;    // per coefficient id: its value as a small integer, or LinearCombination::BIG
    uint32_t intern(const sc &c);
    uint32_t intern_small(int32_t k);    // intern(LinearCombination::scalar_of(k)), without building the scalar for +-1
    // n_mult / n_commit: multipliers allocated and values committed so far; a term outside them throws
    // R1CSError(BPG_EINVAL) (a gadget built from another constraint system's variables).  minus_var: one more
    // term (minus_var, -1) after lc's (the implicit constraint of a multiply: lc - wire = 0).
    void push(const LinearCombination &lc, size_t n_mult, size_t n_commit, const Variable *minus_var = nullptr);

    // ---- following a cached circuit (capture of a circuit the library has seen before).
    // The reference's callers capture one circuit over and over (every *_roundtrip_helper, ref:
    // tests/edwards_point.rs:84-95; a prover service proves one statement shape).  While `follow_key` is set, the
    // three term lists above stay empty and stand for the first `cur` terms of *follow_key: push() only COMPARES the
    // incoming terms with the followed circuit's -- variable, constraint boundaries, and the coefficient VALUE
    // against this store's own dictionary, exactly, no hashing -- and writes nothing.  At the first difference the
    // prefix is copied out (materialize) and capture goes on as usual, so the result never depends on what was
    // followed.  A circuit followed to its end is the cached circuit: prove / verify then take its resident
    // flattening plan without hashing or comparing 12 bytes per term again.
    std::shared_ptr<const CircuitKey> follow_key;
    std::weak_ptr<void> follow_entry;            // the cache entry the key belongs to (r1cs.hip: FlatCached), if any.  Weak: the entry's
                                                 // device memory belongs to its context's cache, not to whoever followed it
    size_t cur = 0;
    void follow(std::shared_ptr<const CircuitKey> key, std::shared_ptr<void> entry);   // only before anything is captured
    void materialize();
    bool followed_fully() const {
        return follow_key && cur == follow_key->term_var.size() && num_constraints == follow_key->num_constraints &&
               coeffs.size() == follow_key->num_coeffs;
    }
    size_t nterms() const { return follow_key ? cur : term_var.size(); }
    std::shared_ptr<const CircuitKey> snapshot(size_t n_mult = 0) const;     // this store's structure as a key (copies)

  private:
    bool push_follow(const LinearCombination &lc, const Variable *minus_var);
};

class ConstraintSystem {
  public:
    virtual ~ConstraintSystem() {}
    struct Triple { Variable l, r, o; };
    virtual Triple multiply(const LinearCombination &left, const LinearCombination &right) = 0;
    // assignment may be null (verifier side / missing witness)
    virtual Triple allocate_multiplier(const Scalar *left, const Scalar *right) = 0;
    virtual void constrain(const LinearCombination &lc) = 0;
    virtual bool is_prover() const = 0;
    // RandomizableConstraintSystem::specify_randomized_constraints / RandomizedConstraintSystem::challenge_scalar
    // (bulletproofs::r1cs, feature yoloproofs; the reference imports RandomizedConstraintSystem at
    // src/gadgets/point/ristretto_point.rs:3 and registers none).  The callback runs inside prove() / verify(),
    // after the first-phase commitments A_I1, A_O1, S1 are in the transcript; it may draw challenges, allocate
    // multipliers (the second phase, committed in A_I2, A_O2, S2) and add constraints.
    using RandomizedFn = std::function<void(ConstraintSystem &)>;
    virtual void specify_randomized_constraints(RandomizedFn cb) = 0;
    // only inside a randomized-constraints callback; throws R1CSError(BPG_EINVAL) elsewhere
    virtual Scalar challenge_scalar(const std::string &label) = 0;
    // what a Variable may refer to: multipliers allocated and values committed so far
    virtual size_t num_multipliers() const = 0;
    virtual size_t num_committed() const = 0;
    virtual const ConstraintStore &store() const = 0;
};

// ---- generators ----------------------------------------------------------------
// PedersenGens::default() and BulletproofGens::new(cap, 1), resident on one GPU:
// table = [ G_0..G_{cap-1} | H_0..H_{cap-1} | B | B_blinding ].  Immutable after creation:
// any number of host threads may prove / verify against one Gens concurrently.
//
// Point-range sharding (BASELINE config 4): rank r of `world` holds only the
// generators lo <= i < hi of G and of H (plus B, B_blinding, which rank 0 alone puts into
// MSMs).  Every rank runs the same prover / verifier on the same inputs; each MSM is computed
// over the rank's range and the 128-byte partial sums are all-gathered through `exchange`
// and added, so every rank sees the same compressed point and the same transcript.
typedef int (*exchange_fn)(void *user, const uint8_t *mine, size_t bytes, uint8_t *all /* world * bytes, rank-major */);
struct Gens {
    bpg_ctx *ctx = nullptr;
    bpg_points *table = nullptr;
    void *d_comb = nullptr;           // radix-16 comb tables of B and B_blinding (Pedersen kernel)
    size_t capacity = 0;
    size_t lo = 0, hi = 0;            // generator index range resident here ([0, capacity) when unsharded)
    int rank = 0, world = 1;
    exchange_fn exchange = nullptr;
    void *exchange_user = nullptr;
    // fixed-base MSM: the table holds tab_blocks blocks of npoints() records, block w = 2^(tab_c w) * block 0.  The MSMs
    // of commitments, inner-product rounds and verification read windows of fb_c bits; where a prover can come to
    // fold its generators (unsharded, capacity >= 4096) the table also holds the half windows (tab_c = fb_c / 2, twice
    // the blocks): the fold then spends one addition per 7-bit digit instead of three per 14-bit digit (msm_fold_submit),
    // and the other MSMs step over every other block (fb_stride).
    int fb_c = 0, fb_windows = 0;
    int tab_c = 0, tab_blocks = 0;
    size_t fb_stride() const { return npoints() * (size_t)(fb_c / tab_c); }
    size_t local() const { return hi - lo; }
    size_t npoints() const { return 2 * local() + 2; }
    size_t off_G() const { return 0; }              // record of G_lo
    size_t off_H() const { return local(); }        // record of H_lo
    size_t off_B() const { return 2 * local(); }
    size_t off_Bb() const { return 2 * local() + 1; }
};
int gens_create(bpg_ctx *ctx, size_t capacity, Gens **out, int rank = 0, int world = 1, exchange_fn exchange = nullptr,
                void *exchange_user = nullptr);
void gens_destroy(Gens *g);

// ---- proof ----------------------------------------------------------------------
struct R1CSProof {
    uint8_t A_I1[32], A_O1[32], S1[32], A_I2[32], A_O2[32], S2[32], T_1[32], T_3[32], T_4[32], T_5[32], T_6[32];
    Scalar t_x, t_x_blinding, e_blinding;
    std::vector<uint8_t> L_vec, R_vec;     // lg n x 32 B each
    Scalar a, b;
    std::vector<uint8_t> to_bytes() const;
    static bool from_bytes(const uint8_t *data, size_t len, R1CSProof &out);
};

// external randomness (the reference passes &mut thread_rng(), ref: tests/edwards_point.rs:79)
typedef void (*rng_fill_fn)(void *user, uint8_t *out, size_t n);

// The blinding draws of ONE large proof, started early (Prover::begin_blindings): a helper thread runs the prover's
// transcript RNG -- one Keccak permutation per 64-byte draw, strictly sequential: 0.42 s of a 0.67 s proof at 2^20
// multipliers -- while the caller's thread is still capturing the circuit, and prove() then uploads the draws in chunks as
// they appear.  The stream does not depend on the circuit, only its length does (3 + 2 n draws), so the helper may
// always run up to twice the multipliers captured so far.
struct BlindingSpec {
    std::thread th;
    std::atomic<size_t> limit{0};         // raw draws the helper may produce: 2 x the multipliers captured so far; 2 n once final
    std::atomic<size_t> done{0};          // raw draws delivered (64-byte slots of `raw`, s_L then s_R)
    std::atomic<bool> heads{false};       // i_bl, o_bl, s_bl drawn
    std::atomic<bool> final_{false}, cancel{false};
    bool failed = false;                  // the helper could not get its pinned buffer (read after join)
    uint8_t *raw = nullptr;               // pinned, `slots` x 64 bytes, from the context's pool
    size_t slots = 0;
    sc i_bl, o_bl, s_bl;
    std::unique_ptr<TranscriptRng> rng;   // the prover's RNG while the helper runs; handed back at exactly 3 + 2 n draws
    bpg_ctx *ctx = nullptr;
    double busy_ms = 0;                   // the helper's drawing time
};

class Prover : public ConstraintSystem {
  public:
    Prover(Gens *gens, const void *label, size_t label_len);
    ~Prover() { cancel_blindings(); }
    // Hand over the 32 bytes of external randomness NOW instead of at prove() (which then does not call its rng), and
    // start drawing blinding factors on a helper thread.  For one large proof proved by prove(): call it after the
    // last commit and before the gadget calls.  A later commit, or a registration of randomized constraints, stops
    // the helper and prove() draws as usual, from the same 32 bytes: the proof's bytes are the same either way.
    int begin_blindings(const uint8_t ext32[32]);
    // Prover::commit; V32 receives the compressed commitment
    Variable commit(const Scalar &v, const Scalar &v_blinding, uint8_t V32[32]);
    // batch form: one GPU launch for m commitments, transcript order preserved
    void commit_batch(const Scalar *v, const Scalar *v_blinding, size_t m, uint8_t *V32, Variable *vars);
    Triple multiply(const LinearCombination &left, const LinearCombination &right) override;
    Triple allocate_multiplier(const Scalar *left, const Scalar *right) override;
    void constrain(const LinearCombination &lc) override { cs_.push(lc, a_L_.size(), v_.size()); }
    // capture against a circuit seen before (ConstraintStore::follow); the constructor picks the prover-side circuit
    // this process flattened last, tests name one
    void follow(std::shared_ptr<const CircuitKey> key, std::shared_ptr<void> entry) {
        if (key && key->n_mult > a_L_.capacity()) {          // the followed circuit's size: no regrowth of the witness vectors
            a_L_.reserve(key->n_mult);
            a_R_.reserve(key->n_mult);
            a_O_.reserve(key->n_mult);
        }
        cs_.follow(std::move(key), std::move(entry));
    }
    bool is_prover() const override { return true; }
    // (upstream's types make a registration from inside a callback impossible: the rcs handed to a callback is not
    // randomizable.  Here the handle is the same object, so the second phase refuses it instead of dropping it.)
    void specify_randomized_constraints(RandomizedFn cb) override {
        if (phase2_) throw R1CSError(BPG_EINVAL, "specify_randomized_constraints inside a randomized-constraints callback");
        cancel_blindings();
        deferred_.push_back(std::move(cb));
    }
    Scalar challenge_scalar(const std::string &label) override;
    // Prover::prove(&bp_gens) names the generators at proving time: a prover made against one generator object
    // (whose Pedersen bases every object shares) may prove against another of the same context
    int set_gens(Gens *g);
    Scalar eval(const LinearCombination &lc) const;
    // Prover::prove(bp_gens); returns BPG_OK or a negative BPG_E* code
    int prove(rng_fill_fn rng, void *rng_user, R1CSProof &out);
    // k proofs by one host thread: byte for byte what k prove() calls in this order give (the external RNG is
    // asked in the same order), but the blinding RNGs of up to eight provers of one circuit shape are drawn in
    // lock-step (TranscriptRngLanes8), which is where most of a proof's host time goes.
    static int prove_many(Prover *const *provers, size_t k, rng_fill_fn rng, void *rng_user, R1CSProof *out);
    size_t num_multipliers() const override { return a_L_.size(); }
    size_t num_committed() const override { return v_.size(); }
    const ConstraintStore &store() const override { return cs_; }
    size_t num_constraints() const { return cs_.num_constraints; }
    Transcript &transcript() { return tr_; }
    struct Timing { double rng_ms = 0, commit_ms = 0, poly_ms = 0, ipp_ms = 0, total_ms = 0; } timing;
    // the scalar-vector kernels of the last single prove() under BPG_TIME_KERNELS=1 ("report GB/s for those"):
    // k_zpow, k_flatten, k_poly, k_eval, k_ipp_rows and k_ipp_fold of round 0 -- dispatch time by HIP events, the bytes each
    // has to move (inputs read once, outputs written once) and its multiplications mod l (Montgomery passes of 104 MAC)
    static constexpr int N_TIMED = 6;
    struct KernelTiming { double us[N_TIMED] = {}, bytes[N_TIMED] = {}, montmuls[N_TIMED] = {}; bool valid = false; } ktiming;

  private:
    // prove() in three steps: transcript RNG set-up, blinding draws (the host's long Keccak chain), the rest
    int prove_begin(rng_fill_fn rng, void *rng_user);
    void draw_blindings();
    void draw_blindings_into(uint8_t *pinned_slot);
    int prove_finish(R1CSProof &out);
    static int finish_group(Prover *const *ps, size_t g, R1CSProof *out);
    static int finish_chunk(Prover *const *ps, size_t k, R1CSProof *out);
    // randomized constraints: first-phase commitments, the deferred callbacks, second-phase commitments
    int two_phase_prologue(R1CSProof &out);
    int commit3(size_t lo, size_t hi, const sc bl[3], const sc *sL, const sc *sR, uint8_t out96[96]);
    std::vector<RandomizedFn> deferred_;
    bool phase2_ = false, two_phase_done_ = false;
    bool consumed_ = false;               // a two-phase prove() has run its callbacks: they cannot run again (upstream: prove(self))
    size_t n1_ = 0;                       // multipliers of the first phase (all of them without randomized constraints)
    sc i_bl2_, o_bl2_, s_bl2_;            // second-phase blindings (zero without a second phase)
    Gens *gens_;
    Transcript tr_;
    ConstraintStore cs_;
    std::vector<sc> a_L_, a_R_, a_O_, v_, v_blinding_;
    // between prove_begin and prove_finish
    std::unique_ptr<TranscriptRng> rng_;
    sc i_bl_, o_bl_, s_bl_;
    std::vector<sc> s_L_, s_R_;
    // one-phase proofs of RAW_DRAW_MIN multipliers or more keep the blinding draws as the RNG delivers them -- 64
    // bytes each, s_L then s_R -- and a kernel reduces them mod l (k_reduce512): on the host that reduction is about
    // half of the drawing time once eight Keccak states advance per permutation
    std::unique_ptr<BlindingSpec> spec_;  // begin_blindings: the helper and its draws
    bool ext_seeded_ = false;             // begin_blindings has taken the external randomness: prove() does not ask its rng
    uint8_t ext_[32];
    void cancel_blindings();
    void publish_multipliers() {
        if (spec_) spec_->limit.store(2 * a_L_.size(), std::memory_order_release);
    }
    std::vector<uint8_t> s_raw_;
    uint8_t *s_raw_pin_ = nullptr;        // the same 128 n bytes in the lane's pinned staging instead (not owned; valid from the draw to the upload)
    static constexpr size_t RAW_DRAW_MIN = 256;
    bool raw_draws() const { return deferred_.empty() && a_L_.size() >= RAW_DRAW_MIN; }
    double t_start_ = 0;
};

// scalars of one proof's verification equation (see Verifier::prepare)
struct VerifyRow {
    sc *d_sc = nullptr;
    size_t np = 0;
    sc rho;                      // batching weight of this equation (verify_batch)
    std::vector<sc> tail;
    std::vector<uint8_t> dyn;
};

class Verifier : public ConstraintSystem {
  public:
    Verifier(const void *label, size_t label_len);
    Variable commit(const uint8_t V32[32]);
    Triple multiply(const LinearCombination &left, const LinearCombination &right) override;
    Triple allocate_multiplier(const Scalar *left, const Scalar *right) override;
    void constrain(const LinearCombination &lc) override { cs_.push(lc, num_vars_, V_.size() / 32); }
    void follow(std::shared_ptr<const CircuitKey> key, std::shared_ptr<void> entry) { cs_.follow(std::move(key), std::move(entry)); }
    bool is_prover() const override { return false; }
    void specify_randomized_constraints(RandomizedFn cb) override {
        if (phase2_) throw R1CSError(BPG_EINVAL, "specify_randomized_constraints inside a randomized-constraints callback");
        deferred_.push_back(std::move(cb));
    }
    Scalar challenge_scalar(const std::string &label) override;
    // Verifier::verify(&proof, &pc_gens, &bp_gens, &mut rng); BPG_OK or BPG_EVERIFY / BPG_EGENS / ...
    int verify(const R1CSProof &proof, Gens *gens, rng_fill_fn rng, void *rng_user);
    int prepare(const R1CSProof &proof, Gens *gens, rng_fill_fn rng, void *rng_user, VerifyRow &out);
    size_t num_multipliers() const override { return num_vars_; }
    size_t num_committed() const override { return V_.size() / 32; }
    const ConstraintStore &store() const override { return cs_; }
    size_t num_constraints() const { return cs_.num_constraints; }

  private:
    Transcript tr_;
    ConstraintStore cs_;
    size_t num_vars_ = 0;
    std::vector<uint8_t> V_;
    std::vector<RandomizedFn> deferred_;
    bool phase2_ = false;
};

// k proofs, one combined multiscalar multiplication (BASELINE config 5); BPG_OK iff all verify
int verify_batch(Verifier *const *vs, const R1CSProof *proofs, size_t k, Gens *gens, rng_fill_fn rng, void *rng_user);

// ---- zerocaf "Sonny" curve over F_l (witness values only) -------------------------
struct SonnyPoint {
    Scalar X, Y, Z, T;
    static SonnyPoint identity() { return {Scalar(0), Scalar(1), Scalar(1), Scalar(0)}; }
};
Scalar sonny_a();     // zerocaf::constants::EDWARDS_A  (= -1)
Scalar sonny_d();     // zerocaf::constants::EDWARDS_D  (= -86650/86649)
SonnyPoint sonny_add(const SonnyPoint &p, const SonnyPoint &q);   // add-2008-hwcd, the formula the gadgets arithmetise
inline SonnyPoint sonny_double(const SonnyPoint &p) { return sonny_add(p, p); }

}  // namespace bpg
