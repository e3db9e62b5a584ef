/*
 * bpg.h -- C ABI of the MI355X-native Bulletproofs R1CS proving backend.
 *
 * This is the drop-in boundary for the hot path that dusk-network/bulletproofs_gadgets
 * drives through its Rust dependencies.  The reference has no FFI of its own (it
 * is a plain Rust library linked against `bulletproofs` / `curve25519-dalek`), so
 * each entry point below names the Rust call it stands in for and the reference
 * call site that reaches it.  Plain pointers and sizes only; every function
 * returns 0 or a negative BPG_E* code; no exceptions cross the boundary.
 *
 * All scalars are 32-byte little-endian integers reduced mod
 * l = 2^252 + 27742317777372353535851937790883648493 (dalek `Scalar::as_bytes`);
 * all points are 32-byte ristretto255 encodings (dalek `CompressedRistretto`).
 *
 * Threading: one bpg_ctx per host thread / GPU.  Calls on one ctx are not
 * re-entrant.  Resident tables (bpg_points) are immutable after creation.
 */
#ifndef BPG_H
#define BPG_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes.  The negative R1CS codes mirror bulletproofs::r1cs::R1CSError
 * (the reference names R1CSError::VerificationError at tests/edwards_point.rs:346). */
#define BPG_OK 0
#define BPG_EINVAL (-1)               /* bad argument */
#define BPG_EDEVICE (-2)              /* HIP runtime failure; see bpg_last_error() */
#define BPG_EFORMAT (-3)              /* R1CSError::FormatError / invalid point encoding */
#define BPG_EVERIFY (-4)              /* R1CSError::VerificationError */
#define BPG_EGENS (-5)                /* R1CSError::InvalidGeneratorsLength */
#define BPG_EMISSING (-6)             /* R1CSError::MissingAssignment */
#define BPG_EGADGET (-7)              /* R1CSError::GadgetError */
#define BPG_ENOMEM (-8)

typedef struct bpg_ctx bpg_ctx;
typedef struct bpg_points bpg_points;   /* resident table of points in HBM (affine Niels form) */
typedef struct bpg_msm_job bpg_msm_job; /* one in-flight MSM */

/* ---- context ---- */
/* Binds to HIP device `device_id` (one process / ctx per GPU).  Fails with
 * BPG_EDEVICE when no MI355X-class device is present: there is no CPU fallback. */
int bpg_init(int device_id, bpg_ctx **out);
void bpg_shutdown(bpg_ctx *ctx);
const char *bpg_last_error(void);
const char *bpg_version(void);

/* ---- device memory helpers (callers that already hold device pointers, e.g. a
 * torch tensor's data_ptr(), can pass those instead) ---- */
int bpg_device_alloc(bpg_ctx *ctx, size_t bytes, void **d_ptr);
int bpg_device_free(bpg_ctx *ctx, void *d_ptr);
int bpg_copy_to_device(bpg_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int bpg_copy_to_host(bpg_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int bpg_synchronize(bpg_ctx *ctx);

/* ---- resident point tables ----
 * Replaces: CompressedRistretto::decompress over a slice (dalek), reached from
 * Verifier::verify (ref: tests/edwards_point.rs:79) and from any caller that holds
 * RistrettoPoint bases for RistrettoPoint::vartime_multiscalar_mul.
 * Returns BPG_EFORMAT if any encoding is invalid (dalek: decompress() == None). */
int bpg_points_from_compressed(bpg_ctx *ctx, const uint8_t *points32, size_t n, bpg_points **out);
/* Replaces: RistrettoPoint::from_uniform_bytes per 64-byte chunk (dalek), the map
 * bulletproofs' GeneratorsChain applies to its SHAKE256 stream
 * (BulletproofGens::new, ref: tests/edwards_point.rs:90). */
int bpg_points_from_uniform(bpg_ctx *ctx, const uint8_t *uniform64, size_t n, bpg_points **out);
/* RistrettoPoint::compress over table entries [start, start+n) (parity / export). */
int bpg_points_compress(bpg_ctx *ctx, const bpg_points *pts, size_t start, size_t n, uint8_t *out32);
size_t bpg_points_len(const bpg_points *pts);
void bpg_points_free(bpg_points *pts);

/* ---- multiscalar multiplication ----
 * Replaces: RistrettoPoint::multiscalar_mul / vartime_multiscalar_mul /
 * optional_multiscalar_mul (dalek Straus / Pippenger), the calls under
 * Prover::prove (A_I, A_O, S, IPP L/R) and Verifier::verify's combined check
 * (ref call sites: tests/edwards_point.rs:47,79,134,158,215,242,320,345,391,412;
 * src/gadgets/scalar.rs:46,63; src/gadgets/sk_knowledge/sk_know_gadget.rs:63,86).
 *
 * out32 = compress( sum_i scalars[i] * pts[i] ), i in [0, n).  n may be 0
 * (identity encoding).  Host-scalar form: copies scalars in, runs, waits. */
int bpg_msm(bpg_ctx *ctx, const bpg_points *pts, const uint8_t *scalars32, size_t n, uint8_t out32[32]);

/* Device-scalar, multi-target, asynchronous form.  d_scalars holds n_targets
 * rows of n scalars (row t at byte offset t*n*32), already resident in HBM.
 * Target t is sum_i d_scalars[t][i] * pts[i].  Several jobs may be in flight;
 * bpg_msm_wait() finishes one (window combination + compress on the host) and
 * writes n_targets*32 bytes. */
int bpg_msm_submit(bpg_ctx *ctx, const bpg_points *pts, const void *d_scalars, size_t n, size_t n_targets,
                   bpg_msm_job **job);
int bpg_msm_wait(bpg_msm_job *job, uint8_t *out32);

/* The entry point that allocates nothing.  No counterpart in the reference: it is for callers that cannot call
 * hipMalloc -- tight serving loops, harnesses that forbid allocation, code under HIP graph capture (only kernel launches and
 * hipMemsetAsync on the caller's stream; not tested under capture here).  Same result as
 * bpg_msm_submit + bpg_msm_wait with one target: d_out32[0..32) = compress( sum_i d_scalars[i] * pts[i] ), i < n.
 * d_scalars and d_out32 are DEVICE pointers, `stream` a hipStream_t (NULL: the null stream); every kernel is enqueued
 * on that stream and the call returns without synchronising: the result is there when the stream reaches that point.
 * No hipMalloc / hipFree / stream / event creation, no host buffer: the kernels' workspace is a static device array of
 * 20 MiB inside the library (each GPU the library runs on holds it until the process ends), which limits
 * n <= BPG_MSM_NOALLOC_MAX_POINTS; beyond that BPG_EINVAL (use bpg_msm_submit).  All calls on one GPU share that
 * array: calls that overlap on one GPU (two streams, two threads, two contexts on the same device) race; issue them
 * on one stream or order them yourself.  The window combination that bpg_msm_wait does on a host core runs on one
 * GPU lane here (about a millisecond): the variant trades latency for not allocating. */
#define BPG_MSM_NOALLOC_MAX_POINTS 16384
int bpg_msm_noalloc(bpg_ctx *ctx, const bpg_points *pts, const void *d_scalars, size_t n, void *d_out32, void *stream);

/* Point-range sharding (one large MSM split over the GPUs of a node):
 * each rank runs its range with bpg_msm_submit(), takes its partial sums with
 * bpg_msm_wait_partial() (n_targets x 128 B: extended coordinates X|Y|Z|T,
 * 32-byte little-endian each, not normalised), the ranks exchange them
 * (an all-gather of 128 B per rank over RCCL), and bpg_msm_combine_partials()
 * adds the k partials and compresses.  Replaces nothing in dalek (its MSM is
 * single-threaded); the combined result equals the unsharded MSM's encoding.
 * combine returns BPG_EFORMAT if a partial is not a point of the curve. */
int bpg_msm_wait_partial(bpg_msm_job *job, uint8_t *out128);
int bpg_msm_combine_partials(const uint8_t *partials128, size_t k, uint8_t out32[32]);

/* Same, but the scalar row addresses table entries through up to 4 segments:
 * scalar i in segment k (segments concatenated) multiplies table entry
 * seg_start[k] + (i - sum of earlier seg_len).  This is how <a,G[0..n)> +
 * <b,H[0..n)> + c*B_blinding is expressed over one generator table. */
int bpg_msm_submit_segments(bpg_ctx *ctx, const bpg_points *pts, const size_t *seg_start, const size_t *seg_len,
                            size_t n_segments, const void *d_scalars, size_t n_targets, bpg_msm_job **job);

/* Diagnostics for bench.py: geometry the planner chose for (n, n_targets). */
int bpg_msm_plan(bpg_ctx *ctx, size_t n, size_t n_targets, int *window_bits, int *windows, int *buckets,
                 int *threads_per_bucket);
/* Test hook, no counterpart in the reference: the k-th device allocation from now on (1 = the next one) made by a
 * growable workspace of this library fails as if the GPU were out of memory; 0 disarms.  Lets the tests walk the
 * failure paths of bpg_msm_submit / prove / verify without exhausting 288 GB. */
void bpg_debug_fail_alloc(int countdown);

/* Per-launch device time in microseconds of the bucket-accumulation kernel of
 * the most recently *waited* job, measured with HIP events on the job's stream. */
int bpg_msm_last_kernel_us(bpg_ctx *ctx, float *accumulate_us, float *total_device_us);

/* ======================================================================================
 * R1CS layer: the bulletproofs::r1cs API the reference's gadgets and tests call.
 *
 * Variables cross the ABI as uint64 = kind << 32 | index with kind 0 = Committed,
 * 1 = MultiplierLeft, 2 = MultiplierRight, 3 = MultiplierOutput, 4 = One
 * (bulletproofs::r1cs::Variable).  A LinearCombination crosses as parallel arrays
 * (vars[n], coeffs[n][32]).  External randomness is injected through bpg_rng_fn,
 * as the reference injects `&mut thread_rng()` (ref: tests/edwards_point.rs:31,79).
 * ====================================================================================== */
typedef struct bpg_gens bpg_gens;               /* PedersenGens::default() + BulletproofGens::new(cap, 1), resident in HBM */
typedef struct bpg_prover bpg_prover;           /* r1cs::Prover (owns its Transcript)  */
typedef struct bpg_verifier bpg_verifier;       /* r1cs::Verifier (owns its Transcript) */
typedef struct bpg_cs bpg_cs;                   /* &mut dyn ConstraintSystem: a Prover or a Verifier */
typedef struct bpg_point_gadget bpg_point_gadget; /* Sonny{Edwards,Ristretto}PointGadget {X, Y, Z, T: LC} */
typedef void (*bpg_rng_fn)(void *user, uint8_t *out, size_t n);

/* BulletproofGens::new(gens_capacity, 1) + PedersenGens::default()
 * (ref: tests/edwards_point.rs:89-90,167-168,252-253,285-286,366-367; scalar.rs:69-70; sk_know_gadget.rs:94-95).
 * party_capacity must be 1, the only value the reference passes (R1CS proofs use share(0)); else BPG_EINVAL.
 * The object is immutable once created: any number of host threads may prove and verify
 * against it at the same time, each on its own stream (one proof = one thread, as in the reference). */
int bpg_gens_create(bpg_ctx *ctx, size_t gens_capacity, size_t party_capacity, bpg_gens **out);
/* Point-range sharding of ONE proof over the GPUs of a node (BASELINE config 4).  No counterpart in
 * the reference (its prover is single-threaded).  Rank `rank` of `world` keeps the generators
 * [cap*rank/world, cap*(rank+1)/world) of G and of H resident.  Every rank then runs the SAME Prover / Verifier
 * calls on the SAME inputs (circuit, witness, external randomness); inside prove / verify each multiscalar
 * multiplication runs over the rank's range only and `exchange` all-gathers the 128-byte partial sums
 * (mine: `bytes` bytes in; all: world * bytes out, rank-major; return 0 on success), e.g. an RCCL all-gather.
 * The partials are added on every rank, so all ranks see identical points, transcripts, proof bytes and
 * verdicts -- the bytes an unsharded bpg_gens produces.  One proof at a time per sharded object
 * (the exchange is a collective). */
typedef int (*bpg_exchange_fn)(void *user, const uint8_t *mine, size_t bytes, uint8_t *all);
int bpg_gens_create_sharded(bpg_ctx *ctx, size_t gens_capacity, size_t party_capacity, int rank, int world,
                            bpg_exchange_fn exchange, void *exchange_user, bpg_gens **out);
void bpg_gens_free(bpg_gens *gens);
size_t bpg_gens_capacity(const bpg_gens *gens);
/* compress generators: which = 'G' | 'H' (party-0 share; global indices, BPG_EINVAL outside a sharded object's range),
 * 'B' (pc_gens.B), 'b' (pc_gens.B_blinding) */
int bpg_gens_export(bpg_ctx *ctx, const bpg_gens *gens, int which, size_t start, size_t n, uint8_t *out32);

/* Transcript::new(label) + Prover::new(&pc_gens, &mut transcript)   ref: tests/edwards_point.rs:25-28 */
int bpg_prover_new(bpg_gens *gens, const uint8_t *label, size_t label_len, bpg_prover **out);
void bpg_prover_free(bpg_prover *p);
bpg_cs *bpg_prover_cs(bpg_prover *p);
/* Prover::commit(v, v_blinding) -> (CompressedRistretto, Variable)   ref: edwards_point.rs:267; tests/edwards_point.rs:31 */
int bpg_prover_commit(bpg_prover *p, const uint8_t v32[32], const uint8_t blinding32[32], uint8_t V32[32], uint64_t *var);
/* m commits in transcript order with one kernel launch (ref: sk_know_gadget.rs:52-55 commits 256 bits one by one) */
int bpg_prover_commit_batch(bpg_prover *p, const uint8_t *v32, const uint8_t *blinding32, size_t m, uint8_t *V32, uint64_t *vars);
/* The generators of the next bpg_prover_prove: Prover::new takes only the Pedersen bases and Prover::prove(&bp_gens)
 * names the generator vectors (ref: tests/edwards_point.rs:28 vs :47).  Every bpg_gens of one context carries the
 * same Pedersen bases, so a prover created against one (say the capacity-1 object behind PedersenGens::default())
 * can prove against another.  BPG_EINVAL if `gens` belongs to another context. */
int bpg_prover_set_gens(bpg_prover *p, bpg_gens *gens);
/* Prover::prove(&bp_gens) -> R1CSProof::to_bytes()   ref: tests/edwards_point.rs:47,134,215,320,391; scalar.rs:46; sk_know_gadget.rs:63
 * Returns BPG_EGENS for R1CSError::InvalidGeneratorsLength.  *proof_len is set even when cap is too small (BPG_EINVAL). */
int bpg_prover_prove(bpg_prover *p, bpg_rng_fn rng, void *rng_user, uint8_t *proof_out, size_t cap, size_t *proof_len);
/* k independent proofs by the calling thread: the same bytes as k bpg_prover_prove calls in this order with the
 * same `rng` (which is asked for 32 bytes per proof, in order), i.e. the loop a caller of the reference writes
 * around Prover::prove (ref: tests/edwards_point.rs:47,134,...: one prove per statement).  The blinding RNGs of up
 * to eight provers of one circuit shape advance in lock-step (eight Keccak states per AVX-512 permutation), which
 * takes most of the host time out of a proof.  Proof j is written at proofs_out + j * cap_each, its length to
 * proof_lens[j].  BPG_EINVAL if a proof does not fit cap_each (proof_lens is still set). */
int bpg_prover_prove_many(bpg_prover *const *provers, size_t k, bpg_rng_fn rng, void *rng_user, uint8_t *proofs_out, size_t cap_each,
                          size_t *proof_lens);
/* One LARGE proof (ref call site: sk_know_gadget.rs:63, Prover::prove after 256 commits and the scalar_mul gadget; BASELINE
 * config 4 = 2^20 multipliers): the prover's blinding vectors s_L, s_R come from one transcript RNG, one Keccak
 * permutation per 64-byte draw, strictly sequential -- 0.42 s on one host core at 2^20, more than capture and GPU work
 * together.  Upstream's prove() asks thread_rng() for 32 bytes and starts that chain at the top of prove(); this call
 * hands the same 32 bytes over EARLIER -- after the last commit, before the gadget calls -- so that a helper thread draws
 * while the caller captures the circuit; bpg_prover_prove then does not call its rng and uploads the draws as they appear.
 * The proof is byte for byte the one bpg_prover_prove gives with an rng that returns ext32.  A commit or a
 * specify_randomized_constraints after this call stops the helper (prove draws as usual, still from ext32).
 * No counterpart call in the reference (its RNG is internal); optional. */
int bpg_prover_begin_blindings(bpg_prover *p, const uint8_t ext32[32]);
int bpg_prover_stats(const bpg_prover *p, size_t *multipliers, size_t *constraints);
/* milliseconds of the last prove(): blinding RNG, A/S commitments, polynomial phase, inner-product argument, total */
int bpg_prover_timing(const bpg_prover *p, double out_ms[5]);

/* Diagnostics for bench.py (HBM GB/s of the scalar-vector kernels under Prover::prove, ref: tests/edwards_point.rs:47):
 * with BPG_TIME_KERNELS=1 in the environment at bpg_init, a single bpg_prover_prove times six of its kernels with HIP
 * events on their dispatches -- k_zpow, k_flatten, k_poly, k_eval, and k_ipp_rows / k_ipp_fold of the first inner-product
 * round -- and keeps, per kernel, the microseconds, the bytes it has to move (every input read once, every output written
 * once) and its multiplications mod l (one Montgomery pass = 104 multiply-adds).  BPG_EINVAL if the last prove was not timed. */
int bpg_prover_kernel_timing(const bpg_prover *p, double out_us[6], double out_bytes[6], double out_montmuls[6]);

/* Transcript::new(label) + Verifier::new(&mut transcript)   ref: tests/edwards_point.rs:59-61 */
int bpg_verifier_new(const uint8_t *label, size_t label_len, bpg_verifier **out);
void bpg_verifier_free(bpg_verifier *v);
bpg_cs *bpg_verifier_cs(bpg_verifier *v);
/* Verifier::commit(V) -> Variable   ref: edwards_point.rs:283; tests/edwards_point.rs:63 */
int bpg_verifier_commit(bpg_verifier *v, const uint8_t V32[32], uint64_t *var);
/* m commits in order (ref: sk_know_gadget.rs:78-81 commits 256 bit commitments one by one) */
int bpg_verifier_commit_batch(bpg_verifier *v, const uint8_t *V32, size_t m, uint64_t *vars);
/* Verifier::verify(&proof, &pc_gens, &bp_gens, &mut rng)   ref: tests/edwards_point.rs:79,158,242,345,412; scalar.rs:63; sk_know_gadget.rs:86
 * BPG_OK | BPG_EVERIFY (R1CSError::VerificationError) | BPG_EFORMAT | BPG_EGENS */
int bpg_verifier_verify(bpg_verifier *v, bpg_gens *gens, const uint8_t *proof, size_t proof_len, bpg_rng_fn rng, void *rng_user);
/* k proofs checked by ONE combined multiscalar multiplication: sum_j rho_j * (verification equation of proof j)
 * (BASELINE config 5; SURVEY 8b bpg_r1cs_verify_batch).  `rng` is asked for 32 bytes per proof, in order, exactly as k
 * bpg_verifier_verify calls would ask; rho_j is the second draw (after r) of verifier j's transcript RNG, so it is
 * bound to proof j and to those 32 bytes and is never zero (BPG_EINVAL if it ever were).  Each verifier carries its
 * own circuit and commitments, as for bpg_verifier_verify, and is consumed the same way.  BPG_OK iff every proof
 * verifies; BPG_EVERIFY says at least one does not (verify them one by one to find which). */
int bpg_verifier_verify_batch(bpg_verifier *const *verifiers, const uint8_t *const *proofs, const size_t *proof_lens, size_t k,
                              bpg_gens *gens, bpg_rng_fn rng, void *rng_user);
int bpg_verifier_stats(const bpg_verifier *v, size_t *multipliers, size_t *constraints);

/* ConstraintSystem::multiply / allocate_multiplier / constrain
 * (ref: edwards_point.rs:45-85; scalar.rs:10-21; boolean.rs:16,20).  out = (left, right, output) variables.
 * allocate_multiplier with NULL assignments on a prover returns BPG_EMISSING (R1CSError::MissingAssignment).
 * Every variable of a linear combination is checked: BPG_EINVAL for an unknown kind, a multiplier index that this
 * constraint system has not allocated, or a commitment index it has not committed (a variable of another prover,
 * a stale handle); the Rust type system rules these out for the reference, its Prover panics on the index. */
int bpg_cs_multiply(bpg_cs *cs, const uint64_t *lvars, const uint8_t *lcoeffs, size_t nl, const uint64_t *rvars,
                    const uint8_t *rcoeffs, size_t nr, uint64_t out[3]);
int bpg_cs_allocate_multiplier(bpg_cs *cs, const uint8_t *left32, const uint8_t *right32, uint64_t out[3]);
int bpg_cs_constrain(bpg_cs *cs, const uint64_t *vars, const uint8_t *coeffs, size_t n);

/* RandomizableConstraintSystem::specify_randomized_constraints(|rcs| { ... }) and
 * RandomizedConstraintSystem::challenge_scalar(label) -- bulletproofs::r1cs with feature yoloproofs (ref: Cargo.toml:18);
 * the reference imports RandomizedConstraintSystem at src/gadgets/point/ristretto_point.rs:3 and none of its gadgets
 * registers a randomized constraint, so nothing in the reference exercises this pair.  `callback` runs
 * later, inside bpg_prover_prove / bpg_verifier_verify, once the first-phase commitments A_I1, A_O1, S1 are in the
 * transcript; `rcs` is valid only during that call and accepts bpg_cs_challenge_scalar, bpg_cs_multiply,
 * bpg_cs_allocate_multiplier, bpg_cs_constrain and the gadget functions.  Multipliers allocated there form the second
 * phase (A_I2, A_O2, S2; proof version byte 1).  The callback returns 0 or a BPG_E* code, which prove / verify then
 * return.  bpg_cs_challenge_scalar outside a callback is BPG_EINVAL.  Provers with randomized constraints passed to
 * bpg_prover_prove_many are proved one at a time. */
typedef int (*bpg_randomized_fn)(void *user, bpg_cs *rcs);
int bpg_cs_specify_randomized_constraints(bpg_cs *cs, bpg_randomized_fn callback, void *user);
int bpg_cs_challenge_scalar(bpg_cs *rcs, const uint8_t *label, size_t label_len, uint8_t out32[32]);

/* The library keeps the flattened structure of the last few circuits it has proved or verified resident on the GPU
 * (which variable, coefficient and constraint each term belongs to, sorted by variable), so repeated proofs of one
 * gadget -- what every *_roundtrip_helper of the reference does, ref: tests/edwards_point.rs:84-95 -- skip that
 * work after the first.  Counters since bpg_init: lookups that found the circuit / that had to build it. */
int bpg_flat_cache_stats(bpg_ctx *ctx, uint64_t *hits, uint64_t *misses);

/* The constraint system as captured so far, one record per linear-combination term in capture order:
 * vars[t] (packed as above), coeffs32[t] (32 bytes), constraint[t] (index of the constrain() call, counting the two
 * implicit ones of each multiply()).  *n_terms is always set; BPG_EINVAL if cap is too small.  No counterpart in
 * bulletproofs (its constraint vector is private): for tests that pin the wiring of a gadget against the
 * reference's source, and for callers that cache per-circuit data. */
int bpg_cs_export_terms(const bpg_cs *cs, uint64_t *vars, uint8_t *coeffs32, uint32_t *constraint, size_t cap, size_t *n_terms);

/* The reference's gadget library (same circuits, call for call). */
int bpg_gadget_binary_constrain(bpg_cs *cs, uint64_t bit_var);                      /* ref: src/gadgets/boolean.rs:12-21 */
/* ref: src/gadgets/scalar.rs:9-26; BPG_EGADGET where the reference panics (zero assignment, :13-15) */
int bpg_gadget_nonzero(bpg_cs *cs, const uint64_t *vars, const uint8_t *coeffs, size_t n, const uint8_t *assignment32);
/* points cross as X | Y | Z | T, 32-byte little-endian Sonny field elements (zerocaf FieldElement::to_bytes) */
int bpg_point_from_coords(const uint8_t xyzt128[128], bpg_point_gadget **out);      /* SonnyEdwardsPointGadget::from_point, edwards_point.rs:19-26 */
int bpg_point_from_vars(const uint64_t vars[4], bpg_point_gadget **out);
void bpg_point_free(bpg_point_gadget *p);
int bpg_edwards_add(bpg_cs *cs, const bpg_point_gadget *a, const bpg_point_gadget *b, bpg_point_gadget **out);   /* edwards_point.rs:28-93 */
int bpg_edwards_double(bpg_cs *cs, const bpg_point_gadget *a, bpg_point_gadget **out);                           /* :97-137 */
int bpg_edwards_equal(bpg_cs *cs, const bpg_point_gadget *a, const bpg_point_gadget *b);                         /* :166-174 */
int bpg_edwards_satisfy_curve_eq(bpg_cs *cs, const bpg_point_gadget *a);                                         /* :178-211 */
int bpg_edwards_conditionally_select(bpg_cs *cs, const bpg_point_gadget *a, const uint64_t *bit_vars, const uint8_t *bit_coeffs,
                                     size_t n, bpg_point_gadget **out);                                          /* :215-253 */
int bpg_edwards_scalar_mul(bpg_cs *cs, const bpg_point_gadget *point, const uint64_t *sk_vars, size_t nbits,
                           bpg_point_gadget **out);                                                              /* :140-163 */
int bpg_edwards_prover_commit(bpg_prover *p, const uint8_t xyzt128[128], const uint8_t blindings128[128], bpg_point_gadget **out,
                              uint8_t V128[128]);                                                                /* :255-276 */
/* BPG_EGADGET unless n_commitments == 4 (the reference asserts, :282) */
int bpg_edwards_verifier_commit(bpg_verifier *v, const uint8_t *V32, size_t n_commitments, bpg_point_gadget **out); /* :278-290 */
int bpg_ristretto_from_point(bpg_cs *cs, const uint8_t xyzt128[128], bpg_point_gadget **out);                    /* ristretto_point.rs:23-32 */
/* BPG_EGADGET unless n_lcs == 4 (the reference asserts, ristretto_point.rs:35) */
int bpg_ristretto_from_lcs(bpg_cs *cs, const bpg_point_gadget *lcs, size_t n_lcs, bpg_point_gadget **out);       /* :34-45 */
int bpg_ristretto_add(bpg_cs *cs, const bpg_point_gadget *a, const bpg_point_gadget *b, bpg_point_gadget **out); /* :92-168 */
int bpg_ristretto_double(bpg_cs *cs, const bpg_point_gadget *a, bpg_point_gadget **out);                         /* :177-180 */
int bpg_ristretto_equals(bpg_cs *cs, const bpg_point_gadget *a, const bpg_point_gadget *b);                      /* :171-175 */
int bpg_ristretto_conditionally_select(bpg_cs *cs, const bpg_point_gadget *a, const uint64_t *bit_vars, const uint8_t *bit_coeffs,
                                       size_t n, bpg_point_gadget **out);                                        /* :184-226 */
int bpg_sk_knowledge_gadget(bpg_cs *cs, const bpg_point_gadget *basep, const bpg_point_gadget *pk, const uint64_t *sk_vars,
                            size_t nbits);                                           /* ref: src/gadgets/sk_knowledge/sk_know_gadget.rs:13-38 */

/* Merlin: Transcript::new(label); append_message(msg_label, msg); challenge_bytes(challenge_label) -> out.
 * Exposed so the Fiat-Shamir layer can be checked against Merlin's published known-answer vector. */
int bpg_transcript_kat(const uint8_t *label, size_t label_len, const char *msg_label, const uint8_t *msg, size_t msg_len,
                       const char *challenge_label, uint8_t *out, size_t out_len);

#ifdef __cplusplus
}
#endif
#endif /* BPG_H */
