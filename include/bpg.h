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
/* Per-launch device time in microseconds of the bucket-accumulation kernel of
 * the most recently *waited* job, measured with HIP events on the job's stream. */
int bpg_msm_last_kernel_us(bpg_ctx *ctx, float *accumulate_us, float *total_device_us);

#ifdef __cplusplus
}
#endif
#endif /* BPG_H */
