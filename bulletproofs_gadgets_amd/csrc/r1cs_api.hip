// r1cs_api.hip -- C ABI of the R1CS layer (include/bpg.h, second half): generators,
// Prover / Verifier, constraint-system primitives with marshalled linear
// combinations, and the reference's gadget library.  C++ exceptions (the reference's
// panics and R1CSError values) are turned into BPG_E* codes here.
#include <cstring>

#include "gadgets.h"
#include "r1cs.h"

using namespace bpg;

struct bpg_gens {
    Gens *g;
};
struct bpg_cs {
    ConstraintSystem *cs;
};
struct bpg_prover {
    Prover *p;
    bpg_cs cs;
};
struct bpg_verifier {
    Verifier *v;
    bpg_cs cs;
};
struct bpg_point_gadget {
    LC X, Y, Z, T;
};

namespace {

Variable unpack_var(uint64_t v) { return Variable{(uint32_t)(v >> 32), (uint32_t)v}; }
uint64_t pack_var(const Variable &v) { return (uint64_t)v.kind << 32 | v.index; }

// Scalar::from_bytes_mod_order
Scalar sc_in(const uint8_t *b) { return Scalar::from_bytes(b); }

// A variable that crosses the ABI is checked against the constraint system it is used with: the kind must be one
// of the five, a multiplier index must have been allocated and a commitment index committed.  (An index past the
// end would write past the flattening plan's slot table and read past the witness vectors.)
Variable checked_var(const ConstraintSystem &cs, uint64_t packed) {
    const Variable v = unpack_var(packed);
    bool ok;
    switch (v.kind) {
        case VAR_COMMITTED: ok = v.index < cs.num_committed(); break;
        case VAR_LEFT:
        case VAR_RIGHT:
        case VAR_OUTPUT: ok = v.index < cs.num_multipliers(); break;
        case VAR_ONE: ok = v.index == 0; break;
        default: ok = false;
    }
    if (!ok) throw R1CSError(BPG_EINVAL, "variable does not belong to this constraint system");
    return v;
}

LC unpack_lc(const ConstraintSystem &cs, const uint64_t *vars, const uint8_t *coeffs, size_t n) {
    if (n && (!vars || !coeffs)) throw R1CSError(BPG_EINVAL, "linear combination: null arrays");
    LC lc;
    lc.reserve(n);
    for (size_t i = 0; i < n; i++) lc.push(checked_var(cs, vars[i]), sc_in(coeffs + 32 * i));
    return lc;
}

void out_triple(uint64_t out[3], const ConstraintSystem::Triple &t) {
    out[0] = pack_var(t.l);
    out[1] = pack_var(t.r);
    out[2] = pack_var(t.o);
}

template <typename F>
int guarded(F f) {
    try {
        return f();
    } catch (const R1CSError &e) {
        set_last_error_msg(e.what());
        return e.code;
    } catch (const std::exception &e) {
        set_last_error_msg(e.what());
        return BPG_EGADGET;
    }
}

SonnyPoint point_in(const uint8_t *xyzt128) {
    return SonnyPoint{sc_in(xyzt128), sc_in(xyzt128 + 32), sc_in(xyzt128 + 64), sc_in(xyzt128 + 96)};
}

}  // namespace

extern "C" {

// ---------------------------------------------------------------- generators
int bpg_gens_create(bpg_ctx *ctx, size_t gens_capacity, size_t party_capacity, bpg_gens **out) {
    if (!out || party_capacity != 1) return BPG_EINVAL;
    Gens *g = nullptr;
    int rc = gens_create(ctx, gens_capacity, &g);
    if (rc) return rc;
    *out = new bpg_gens{g};
    return BPG_OK;
}
int bpg_gens_create_sharded(bpg_ctx *ctx, size_t gens_capacity, size_t party_capacity, int rank, int world,
                            bpg_exchange_fn exchange, void *exchange_user, bpg_gens **out) {
    if (!out || party_capacity != 1) return BPG_EINVAL;
    Gens *g = nullptr;
    int rc = gens_create(ctx, gens_capacity, &g, rank, world, exchange, exchange_user);
    if (rc) return rc;
    *out = new bpg_gens{g};
    return BPG_OK;
}
void bpg_gens_free(bpg_gens *g) {
    if (!g) return;
    gens_destroy(g->g);
    delete g;
}
size_t bpg_gens_capacity(const bpg_gens *g) { return g ? g->g->capacity : 0; }
int bpg_gens_export(bpg_ctx *ctx, const bpg_gens *g, int which, size_t start, size_t n, uint8_t *out32) {
    if (!g) return BPG_EINVAL;
    size_t base, first = 0, lim;
    switch (which) {          // G / H: global generator indices; a sharded object holds [lo, hi) only
        case 'G': base = g->g->off_G(); first = g->g->lo; lim = g->g->hi; break;
        case 'H': base = g->g->off_H(); first = g->g->lo; lim = g->g->hi; break;
        case 'B': base = g->g->off_B(); lim = 1; break;
        case 'b': base = g->g->off_Bb(); lim = 1; break;
        default: return BPG_EINVAL;
    }
    if (start < first || start + n > lim) return BPG_EINVAL;
    return bpg_points_compress(ctx, g->g->table, base + (start - first), n, out32);
}

// ---------------------------------------------------------------- prover
int bpg_prover_new(bpg_gens *gens, const uint8_t *label, size_t label_len, bpg_prover **out) {
    if (!gens || !out) return BPG_EINVAL;
    auto *p = new bpg_prover();
    p->p = new Prover(gens->g, label, label_len);
    p->cs.cs = p->p;
    *out = p;
    return BPG_OK;
}
void bpg_prover_free(bpg_prover *p) {
    if (!p) return;
    delete p->p;
    delete p;
}
bpg_cs *bpg_prover_cs(bpg_prover *p) { return p ? &p->cs : nullptr; }

int bpg_prover_commit(bpg_prover *p, const uint8_t v32[32], const uint8_t blinding32[32], uint8_t V32[32], uint64_t *var) {
    if (!p || !v32 || !blinding32 || !V32 || !var) return BPG_EINVAL;
    return guarded([&] {
        *var = pack_var(p->p->commit(sc_in(v32), sc_in(blinding32), V32));
        return BPG_OK;
    });
}
int bpg_prover_commit_batch(bpg_prover *p, const uint8_t *v32, const uint8_t *blinding32, size_t m, uint8_t *V32, uint64_t *vars) {
    if (!p || (m && (!v32 || !blinding32 || !V32 || !vars))) return BPG_EINVAL;
    return guarded([&] {
        std::vector<Scalar> v(m), b(m);
        std::vector<Variable> vv(m);
        for (size_t i = 0; i < m; i++) {
            v[i] = sc_in(v32 + 32 * i);
            b[i] = sc_in(blinding32 + 32 * i);
        }
        p->p->commit_batch(v.data(), b.data(), m, V32, vv.data());
        for (size_t i = 0; i < m; i++) vars[i] = pack_var(vv[i]);
        return BPG_OK;
    });
}
int bpg_prover_set_gens(bpg_prover *p, bpg_gens *gens) {
    if (!p || !gens) return BPG_EINVAL;
    return guarded([&] { return p->p->set_gens(gens->g); });
}
int bpg_prover_prove(bpg_prover *p, bpg_rng_fn rng, void *rng_user, uint8_t *proof_out, size_t cap, size_t *proof_len) {
    if (!p || !rng || !proof_len) return BPG_EINVAL;
    return guarded([&] {
        R1CSProof proof;
        int rc = p->p->prove(rng, rng_user, proof);
        if (rc) return rc;
        std::vector<uint8_t> b = proof.to_bytes();
        *proof_len = b.size();
        if (!proof_out || cap < b.size()) return BPG_EINVAL;
        memcpy(proof_out, b.data(), b.size());
        return BPG_OK;
    });
}
int bpg_prover_prove_many(bpg_prover *const *provers, size_t k, bpg_rng_fn rng, void *rng_user, uint8_t *proofs_out, size_t cap_each,
                          size_t *proof_lens) {
    if (!rng || !proof_lens || (k && !provers)) return BPG_EINVAL;
    for (size_t j = 0; j < k; j++)
        if (!provers[j]) return BPG_EINVAL;
    return guarded([&] {
        std::vector<Prover *> ps(k);
        for (size_t j = 0; j < k; j++) ps[j] = provers[j]->p;
        std::vector<R1CSProof> proofs(k);
        int rc = Prover::prove_many(ps.data(), k, rng, rng_user, proofs.data());
        if (rc) return rc;
        bool fits = proofs_out != nullptr;
        for (size_t j = 0; j < k; j++) {
            std::vector<uint8_t> b = proofs[j].to_bytes();
            proof_lens[j] = b.size();
            if (proofs_out && b.size() <= cap_each) memcpy(proofs_out + j * cap_each, b.data(), b.size());
            else fits = false;
        }
        return fits ? BPG_OK : BPG_EINVAL;
    });
}
int bpg_prover_stats(const bpg_prover *p, size_t *multipliers, size_t *constraints) {
    if (!p) return BPG_EINVAL;
    if (multipliers) *multipliers = p->p->num_multipliers();
    if (constraints) *constraints = p->p->num_constraints();
    return BPG_OK;
}
int bpg_prover_timing(const bpg_prover *p, double out_ms[5]) {
    if (!p || !out_ms) return BPG_EINVAL;
    const auto &t = p->p->timing;
    out_ms[0] = t.rng_ms; out_ms[1] = t.commit_ms; out_ms[2] = t.poly_ms; out_ms[3] = t.ipp_ms; out_ms[4] = t.total_ms;
    return BPG_OK;
}

int bpg_prover_begin_blindings(bpg_prover *p, const uint8_t ext32[32]) {
    if (!p || !ext32) return BPG_EINVAL;
    return guarded([&] { return p->p->begin_blindings(ext32); });
}

int bpg_prover_kernel_timing(const bpg_prover *p, double out_us[6], double out_bytes[6], double out_montmuls[6]) {
    if (!p || !out_us || !out_bytes || !out_montmuls) return BPG_EINVAL;
    const auto &k = p->p->ktiming;
    if (!k.valid) return BPG_EINVAL;
    for (int i = 0; i < Prover::N_TIMED; i++) {
        out_us[i] = k.us[i];
        out_bytes[i] = k.bytes[i];
        out_montmuls[i] = k.montmuls[i];
    }
    return BPG_OK;
}

// ---------------------------------------------------------------- verifier
int bpg_verifier_new(const uint8_t *label, size_t label_len, bpg_verifier **out) {
    if (!out) return BPG_EINVAL;
    auto *v = new bpg_verifier();
    v->v = new Verifier(label, label_len);
    v->cs.cs = v->v;
    *out = v;
    return BPG_OK;
}
void bpg_verifier_free(bpg_verifier *v) {
    if (!v) return;
    delete v->v;
    delete v;
}
bpg_cs *bpg_verifier_cs(bpg_verifier *v) { return v ? &v->cs : nullptr; }
int bpg_verifier_commit(bpg_verifier *v, const uint8_t V32[32], uint64_t *var) {
    if (!v || !V32 || !var) return BPG_EINVAL;
    *var = pack_var(v->v->commit(V32));
    return BPG_OK;
}
int bpg_verifier_commit_batch(bpg_verifier *v, const uint8_t *V32, size_t m, uint64_t *vars) {
    if (!v || (m && (!V32 || !vars))) return BPG_EINVAL;
    return guarded([&] {
        for (size_t i = 0; i < m; i++) vars[i] = pack_var(v->v->commit(V32 + 32 * i));
        return BPG_OK;
    });
}
int bpg_verifier_verify(bpg_verifier *v, bpg_gens *gens, const uint8_t *proof, size_t proof_len, bpg_rng_fn rng, void *rng_user) {
    if (!v || !gens || !proof || !rng) return BPG_EINVAL;
    return guarded([&] {
        R1CSProof P;
        if (!R1CSProof::from_bytes(proof, proof_len, P)) return BPG_EFORMAT;
        return v->v->verify(P, gens->g, rng, rng_user);
    });
}
int bpg_verifier_verify_batch(bpg_verifier *const *verifiers, const uint8_t *const *proofs, const size_t *proof_lens, size_t k,
                              bpg_gens *gens, bpg_rng_fn rng, void *rng_user) {
    if (!gens || !rng || (k && (!verifiers || !proofs || !proof_lens))) return BPG_EINVAL;
    return guarded([&] {
        std::vector<R1CSProof> P(k);
        std::vector<Verifier *> vs(k);
        for (size_t j = 0; j < k; j++) {
            if (!verifiers[j] || !proofs[j]) return BPG_EINVAL;
            if (!R1CSProof::from_bytes(proofs[j], proof_lens[j], P[j])) return BPG_EFORMAT;
            vs[j] = verifiers[j]->v;
        }
        return verify_batch(vs.data(), P.data(), k, gens->g, rng, rng_user);
    });
}
int bpg_verifier_stats(const bpg_verifier *v, size_t *multipliers, size_t *constraints) {
    if (!v) return BPG_EINVAL;
    if (multipliers) *multipliers = v->v->num_multipliers();
    if (constraints) *constraints = v->v->num_constraints();
    return BPG_OK;
}

// ---------------------------------------------------------------- constraint-system primitives
int bpg_cs_multiply(bpg_cs *cs, const uint64_t *lvars, const uint8_t *lcoeffs, size_t nl, const uint64_t *rvars,
                    const uint8_t *rcoeffs, size_t nr, uint64_t out[3]) {
    if (!cs || !out) return BPG_EINVAL;
    return guarded([&] {
        out_triple(out, cs->cs->multiply(unpack_lc(*cs->cs, lvars, lcoeffs, nl), unpack_lc(*cs->cs, rvars, rcoeffs, nr)));
        return BPG_OK;
    });
}
int bpg_cs_allocate_multiplier(bpg_cs *cs, const uint8_t *left32, const uint8_t *right32, uint64_t out[3]) {
    if (!cs || !out) return BPG_EINVAL;
    return guarded([&] {
        if (left32 && right32) {
            Scalar l = sc_in(left32), r = sc_in(right32);
            out_triple(out, cs->cs->allocate_multiplier(&l, &r));
        } else {
            out_triple(out, cs->cs->allocate_multiplier(nullptr, nullptr));
        }
        return BPG_OK;
    });
}
int bpg_cs_constrain(bpg_cs *cs, const uint64_t *vars, const uint8_t *coeffs, size_t n) {
    if (!cs) return BPG_EINVAL;
    return guarded([&] {
        cs->cs->constrain(unpack_lc(*cs->cs, vars, coeffs, n));
        return BPG_OK;
    });
}

int bpg_cs_specify_randomized_constraints(bpg_cs *cs, bpg_randomized_fn callback, void *user) {
    if (!cs || !callback) return BPG_EINVAL;
    return guarded([&] {
        cs->cs->specify_randomized_constraints([callback, user](ConstraintSystem &rcs) {
            bpg_cs handle{&rcs};
            const int rc = callback(user, &handle);
            if (rc) throw R1CSError(rc, "randomized-constraints callback failed");
        });
        return BPG_OK;
    });
}
int bpg_cs_challenge_scalar(bpg_cs *rcs, const uint8_t *label, size_t label_len, uint8_t out32[32]) {
    if (!rcs || !out32 || (label_len && !label)) return BPG_EINVAL;
    return guarded([&] {
        // merlin labels are byte strings without NUL bytes (Rust `&'static [u8]` literals)
        std::string l((const char *)label, label_len);
        if (l.find('\0') != std::string::npos) return BPG_EINVAL;
        rcs->cs->challenge_scalar(l).to_bytes(out32);
        return BPG_OK;
    });
}

int bpg_flat_cache_stats(bpg_ctx *ctx, uint64_t *hits, uint64_t *misses) {
    if (!ctx || !hits || !misses) return BPG_EINVAL;
    *hits = ctx->flat_hits.load();
    *misses = ctx->flat_misses.load();
    return BPG_OK;
}

int bpg_cs_export_terms(const bpg_cs *cs, uint64_t *vars, uint8_t *coeffs32, uint32_t *constraint, size_t cap, size_t *n_terms) {
    if (!cs || !n_terms) return BPG_EINVAL;
    const ConstraintStore &st = cs->cs->store();
    // a store that follows a cached circuit holds its terms as a prefix of that circuit's lists
    const bool fol = (bool)st.follow_key;
    const uint32_t *tv = fol ? st.follow_key->term_var.data() : st.term_var.data();
    const uint32_t *tc = fol ? st.follow_key->term_coeff.data() : st.term_coeff.data();
    const uint32_t *tq = fol ? st.follow_key->term_q.data() : st.term_q.data();
    const size_t n = st.nterms();
    *n_terms = n;
    if (cap < n || (n && (!vars || !coeffs32 || !constraint))) return BPG_EINVAL;
    for (size_t t = 0; t < n; t++) {
        vars[t] = (uint64_t)(tv[t] >> 28) << 32 | (tv[t] & 0x0fffffffu);
        sc_tobytes(coeffs32 + 32 * t, st.coeffs[tc[t]]);
        constraint[t] = tq[t];
    }
    return BPG_OK;
}

// ---------------------------------------------------------------- gadgets
int bpg_gadget_binary_constrain(bpg_cs *cs, uint64_t bit_var) {
    if (!cs) return BPG_EINVAL;
    return guarded([&] {
        binary_constrain_gadget(*cs->cs, checked_var(*cs->cs, bit_var));
        return BPG_OK;
    });
}
int bpg_gadget_nonzero(bpg_cs *cs, const uint64_t *vars, const uint8_t *coeffs, size_t n, const uint8_t *assignment32) {
    if (!cs) return BPG_EINVAL;
    return guarded([&] {
        LC lc = unpack_lc(*cs->cs, vars, coeffs, n);
        if (assignment32) {
            Scalar a = sc_in(assignment32);
            nonzero_gadget(lc, &a, *cs->cs);
        } else {
            nonzero_gadget(lc, nullptr, *cs->cs);
        }
        return BPG_OK;
    });
}

int bpg_point_from_coords(const uint8_t xyzt128[128], bpg_point_gadget **out) {
    if (!xyzt128 || !out) return BPG_EINVAL;
    SonnyPoint p = point_in(xyzt128);
    *out = new bpg_point_gadget{LC(p.X), LC(p.Y), LC(p.Z), LC(p.T)};
    return BPG_OK;
}
int bpg_point_from_vars(const uint64_t vars[4], bpg_point_gadget **out) {
    if (!vars || !out) return BPG_EINVAL;
    *out = new bpg_point_gadget{LC(unpack_var(vars[0])), LC(unpack_var(vars[1])), LC(unpack_var(vars[2])), LC(unpack_var(vars[3]))};
    return BPG_OK;
}
void bpg_point_free(bpg_point_gadget *p) { delete p; }

#define E_OF(p) SonnyEdwardsPointGadget{(p)->X, (p)->Y, (p)->Z, (p)->T}
#define R_OF(p) SonnyRistrettoPointGadget{(p)->X, (p)->Y, (p)->Z, (p)->T}
#define NEW_FROM(g) new bpg_point_gadget{(g).X, (g).Y, (g).Z, (g).T}

int bpg_edwards_add(bpg_cs *cs, const bpg_point_gadget *a, const bpg_point_gadget *b, bpg_point_gadget **out) {
    if (!cs || !a || !b || !out) return BPG_EINVAL;
    return guarded([&] { auto r = E_OF(a).add(E_OF(b), *cs->cs); *out = NEW_FROM(r); return BPG_OK; });
}
int bpg_edwards_double(bpg_cs *cs, const bpg_point_gadget *a, bpg_point_gadget **out) {
    if (!cs || !a || !out) return BPG_EINVAL;
    return guarded([&] { auto r = E_OF(a).double_(*cs->cs); *out = NEW_FROM(r); return BPG_OK; });
}
int bpg_edwards_equal(bpg_cs *cs, const bpg_point_gadget *a, const bpg_point_gadget *b) {
    if (!cs || !a || !b) return BPG_EINVAL;
    return guarded([&] { E_OF(a).equal(E_OF(b), *cs->cs); return BPG_OK; });
}
int bpg_edwards_satisfy_curve_eq(bpg_cs *cs, const bpg_point_gadget *a) {
    if (!cs || !a) return BPG_EINVAL;
    return guarded([&] { E_OF(a).satisfy_curve_eq(*cs->cs); return BPG_OK; });
}
int bpg_edwards_conditionally_select(bpg_cs *cs, const bpg_point_gadget *a, const uint64_t *bit_vars, const uint8_t *bit_coeffs,
                                     size_t n, bpg_point_gadget **out) {
    if (!cs || !a || !out) return BPG_EINVAL;
    return guarded([&] {
        auto r = E_OF(a).conditionally_select(unpack_lc(*cs->cs, bit_vars, bit_coeffs, n), *cs->cs);
        *out = NEW_FROM(r);
        return BPG_OK;
    });
}
int bpg_edwards_scalar_mul(bpg_cs *cs, const bpg_point_gadget *point, const uint64_t *sk_vars, size_t nbits, bpg_point_gadget **out) {
    if (!cs || !point || !out || (nbits && !sk_vars)) return BPG_EINVAL;
    return guarded([&] {
        HostPhase hp(HostPhase::GADGET);
        std::vector<Variable> sk(nbits);
        for (size_t i = 0; i < nbits; i++) sk[i] = checked_var(*cs->cs, sk_vars[i]);
        auto r = SonnyEdwardsPointGadget::scalar_mul(E_OF(point), sk, *cs->cs);
        *out = NEW_FROM(r);
        return BPG_OK;
    });
}
// ref: edwards_point.rs:255-276; blindings come from the caller (the reference draws them from thread_rng)
int bpg_edwards_prover_commit(bpg_prover *p, const uint8_t xyzt128[128], const uint8_t blindings128[128], bpg_point_gadget **out,
                              uint8_t V128[128]) {
    if (!p || !xyzt128 || !blindings128 || !out || !V128) return BPG_EINVAL;
    return guarded([&] {
        Scalar v[4], b[4];
        Variable vars[4];
        for (int i = 0; i < 4; i++) {
            v[i] = sc_in(xyzt128 + 32 * i);
            b[i] = sc_in(blindings128 + 32 * i);
        }
        p->p->commit_batch(v, b, 4, V128, vars);
        *out = new bpg_point_gadget{LC(vars[0]), LC(vars[1]), LC(vars[2]), LC(vars[3])};
        return BPG_OK;
    });
}
// ref: edwards_point.rs:278-290 (assert_eq!(commitments.len(), 4))
int bpg_edwards_verifier_commit(bpg_verifier *v, const uint8_t *V32, size_t n_commitments, bpg_point_gadget **out) {
    if (!v || !V32 || !out) return BPG_EINVAL;
    if (n_commitments != 4) {
        set_last_error_msg("verifier_commit_to_sonny_edwards_point: expected 4 commitments");
        return BPG_EGADGET;
    }
    Variable vars[4];
    for (int i = 0; i < 4; i++) vars[i] = v->v->commit(V32 + 32 * i);
    *out = new bpg_point_gadget{LC(vars[0]), LC(vars[1]), LC(vars[2]), LC(vars[3])};
    return BPG_OK;
}

int bpg_ristretto_from_point(bpg_cs *cs, const uint8_t xyzt128[128], bpg_point_gadget **out) {
    if (!cs || !xyzt128 || !out) return BPG_EINVAL;
    return guarded([&] {
        auto r = SonnyRistrettoPointGadget::from_point(point_in(xyzt128), *cs->cs);
        *out = NEW_FROM(r);
        return BPG_OK;
    });
}
// ref: ristretto_point.rs:34-45: `lcs` is a list of gadgets' worth of LCs; n_lcs must be 4
int bpg_ristretto_from_lcs(bpg_cs *cs, const bpg_point_gadget *lcs, size_t n_lcs, bpg_point_gadget **out) {
    if (!cs || !lcs || !out) return BPG_EINVAL;
    return guarded([&] {
        std::vector<LC> v{lcs->X, lcs->Y, lcs->Z, lcs->T};
        v.resize(n_lcs);
        auto r = SonnyRistrettoPointGadget::from_lcs(v, *cs->cs);
        *out = NEW_FROM(r);
        return BPG_OK;
    });
}
int bpg_ristretto_add(bpg_cs *cs, const bpg_point_gadget *a, const bpg_point_gadget *b, bpg_point_gadget **out) {
    if (!cs || !a || !b || !out) return BPG_EINVAL;
    return guarded([&] { auto r = R_OF(a).add(*cs->cs, R_OF(b)); *out = NEW_FROM(r); return BPG_OK; });
}
int bpg_ristretto_double(bpg_cs *cs, const bpg_point_gadget *a, bpg_point_gadget **out) {
    if (!cs || !a || !out) return BPG_EINVAL;
    return guarded([&] { auto r = R_OF(a).double_(*cs->cs); *out = NEW_FROM(r); return BPG_OK; });
}
int bpg_ristretto_equals(bpg_cs *cs, const bpg_point_gadget *a, const bpg_point_gadget *b) {
    if (!cs || !a || !b) return BPG_EINVAL;
    return guarded([&] { R_OF(a).equals(*cs->cs, R_OF(b)); return BPG_OK; });
}
int bpg_ristretto_conditionally_select(bpg_cs *cs, const bpg_point_gadget *a, const uint64_t *bit_vars, const uint8_t *bit_coeffs,
                                       size_t n, bpg_point_gadget **out) {
    if (!cs || !a || !out) return BPG_EINVAL;
    return guarded([&] {
        auto r = R_OF(a).conditionally_select(unpack_lc(*cs->cs, bit_vars, bit_coeffs, n), *cs->cs);
        *out = NEW_FROM(r);
        return BPG_OK;
    });
}
int bpg_sk_knowledge_gadget(bpg_cs *cs, const bpg_point_gadget *basep, const bpg_point_gadget *pk, const uint64_t *sk_vars,
                            size_t nbits) {
    if (!cs || !basep || !pk || (nbits && !sk_vars)) return BPG_EINVAL;
    return guarded([&] {
        std::vector<Variable> sk(nbits);
        for (size_t i = 0; i < nbits; i++) sk[i] = checked_var(*cs->cs, sk_vars[i]);
        sk_knowledge_gadget(*cs->cs, R_OF(basep), R_OF(pk), sk);
        return BPG_OK;
    });
}

// ---------------------------------------------------------------- transcript (for parity tests of the Fiat-Shamir layer)
int bpg_transcript_kat(const uint8_t *label, size_t label_len, const char *msg_label, const uint8_t *msg, size_t msg_len,
                       const char *challenge_label, uint8_t *out, size_t out_len) {
    if (!label || !msg_label || !challenge_label || !out) return BPG_EINVAL;
    Transcript t(label, label_len);
    t.append_message(msg_label, msg, msg_len);
    t.challenge_bytes(challenge_label, out, out_len);
    return BPG_OK;
}

}  // extern "C"
