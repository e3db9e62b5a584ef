//! Raw declarations of `include/bpg.h`, one for one.  See that header for what each entry point replaces in the
//! reference (`bulletproofs` / `curve25519-dalek` calls, with the reference call sites).  Not compiled in this
//! repository's image (no Rust toolchain); `tests/test_abi.py` checks names and parameter counts against the header.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_double, c_float, c_int, c_void};

pub const BPG_OK: c_int = 0;
pub const BPG_EINVAL: c_int = -1;
pub const BPG_EDEVICE: c_int = -2;
pub const BPG_EFORMAT: c_int = -3;
pub const BPG_EVERIFY: c_int = -4;
pub const BPG_EGENS: c_int = -5;
pub const BPG_EMISSING: c_int = -6;
pub const BPG_EGADGET: c_int = -7;
pub const BPG_ENOMEM: c_int = -8;
pub const BPG_MSM_NOALLOC_MAX_POINTS: usize = 16384;

macro_rules! opaque {
    ($($name:ident),*) => { $( #[repr(C)] pub struct $name { _private: [u8; 0] } )* };
}
opaque!(bpg_ctx, bpg_points, bpg_msm_job, bpg_gens, bpg_prover, bpg_verifier, bpg_cs, bpg_point_gadget);

pub type bpg_rng_fn = Option<unsafe extern "C" fn(user: *mut c_void, out: *mut u8, n: usize)>;
pub type bpg_exchange_fn = Option<unsafe extern "C" fn(user: *mut c_void, mine: *const u8, bytes: usize, all: *mut u8) -> c_int>;
pub type bpg_randomized_fn = Option<unsafe extern "C" fn(user: *mut c_void, rcs: *mut bpg_cs) -> c_int>;

extern "C" {
    // ---- context
    pub fn bpg_init(device_id: c_int, out: *mut *mut bpg_ctx) -> c_int;
    pub fn bpg_shutdown(ctx: *mut bpg_ctx);
    pub fn bpg_last_error() -> *const c_char;
    pub fn bpg_version() -> *const c_char;
    // ---- device memory helpers
    pub fn bpg_device_alloc(ctx: *mut bpg_ctx, bytes: usize, d_ptr: *mut *mut c_void) -> c_int;
    pub fn bpg_device_free(ctx: *mut bpg_ctx, d_ptr: *mut c_void) -> c_int;
    pub fn bpg_copy_to_device(ctx: *mut bpg_ctx, d_dst: *mut c_void, h_src: *const c_void, bytes: usize) -> c_int;
    pub fn bpg_copy_to_host(ctx: *mut bpg_ctx, h_dst: *mut c_void, d_src: *const c_void, bytes: usize) -> c_int;
    pub fn bpg_synchronize(ctx: *mut bpg_ctx) -> c_int;
    // ---- resident point tables
    pub fn bpg_points_from_compressed(ctx: *mut bpg_ctx, points32: *const u8, n: usize, out: *mut *mut bpg_points) -> c_int;
    pub fn bpg_points_from_uniform(ctx: *mut bpg_ctx, uniform64: *const u8, n: usize, out: *mut *mut bpg_points) -> c_int;
    pub fn bpg_points_compress(ctx: *mut bpg_ctx, pts: *const bpg_points, start: usize, n: usize, out32: *mut u8) -> c_int;
    pub fn bpg_points_len(pts: *const bpg_points) -> usize;
    pub fn bpg_points_free(pts: *mut bpg_points);
    // ---- multiscalar multiplication
    pub fn bpg_msm(ctx: *mut bpg_ctx, pts: *const bpg_points, scalars32: *const u8, n: usize, out32: *mut u8) -> c_int;
    pub fn bpg_msm_submit(ctx: *mut bpg_ctx, pts: *const bpg_points, d_scalars: *const c_void, n: usize, n_targets: usize,
                          job: *mut *mut bpg_msm_job) -> c_int;
    pub fn bpg_msm_wait(job: *mut bpg_msm_job, out32: *mut u8) -> c_int;
    pub fn bpg_msm_noalloc(ctx: *mut bpg_ctx, pts: *const bpg_points, d_scalars: *const c_void, n: usize, d_out32: *mut c_void,
                           stream: *mut c_void) -> c_int;
    pub fn bpg_msm_wait_partial(job: *mut bpg_msm_job, out128: *mut u8) -> c_int;
    pub fn bpg_msm_combine_partials(partials128: *const u8, k: usize, out32: *mut u8) -> c_int;
    pub fn bpg_msm_submit_segments(ctx: *mut bpg_ctx, pts: *const bpg_points, seg_start: *const usize, seg_len: *const usize,
                                   n_segments: usize, d_scalars: *const c_void, n_targets: usize, job: *mut *mut bpg_msm_job) -> c_int;
    pub fn bpg_msm_plan(ctx: *mut bpg_ctx, n: usize, n_targets: usize, window_bits: *mut c_int, windows: *mut c_int,
                        buckets: *mut c_int, threads_per_bucket: *mut c_int) -> c_int;
    pub fn bpg_debug_fail_alloc(countdown: c_int);
    pub fn bpg_msm_last_kernel_us(ctx: *mut bpg_ctx, accumulate_us: *mut c_float, total_device_us: *mut c_float) -> c_int;
    // ---- R1CS layer: generators
    pub fn bpg_gens_create(ctx: *mut bpg_ctx, gens_capacity: usize, party_capacity: usize, out: *mut *mut bpg_gens) -> c_int;
    pub fn bpg_gens_create_sharded(ctx: *mut bpg_ctx, gens_capacity: usize, party_capacity: usize, rank: c_int, world: c_int,
                                   exchange: bpg_exchange_fn, exchange_user: *mut c_void, out: *mut *mut bpg_gens) -> c_int;
    pub fn bpg_gens_free(gens: *mut bpg_gens);
    pub fn bpg_gens_capacity(gens: *const bpg_gens) -> usize;
    pub fn bpg_gens_export(ctx: *mut bpg_ctx, gens: *const bpg_gens, which: c_int, start: usize, n: usize, out32: *mut u8) -> c_int;
    // ---- prover
    pub fn bpg_prover_new(gens: *mut bpg_gens, label: *const u8, label_len: usize, out: *mut *mut bpg_prover) -> c_int;
    pub fn bpg_prover_free(p: *mut bpg_prover);
    pub fn bpg_prover_cs(p: *mut bpg_prover) -> *mut bpg_cs;
    pub fn bpg_prover_commit(p: *mut bpg_prover, v32: *const u8, blinding32: *const u8, V32: *mut u8, var: *mut u64) -> c_int;
    pub fn bpg_prover_commit_batch(p: *mut bpg_prover, v32: *const u8, blinding32: *const u8, m: usize, V32: *mut u8,
                                   vars: *mut u64) -> c_int;
    pub fn bpg_prover_set_gens(p: *mut bpg_prover, gens: *mut bpg_gens) -> c_int;
    pub fn bpg_prover_prove(p: *mut bpg_prover, rng: bpg_rng_fn, rng_user: *mut c_void, proof_out: *mut u8, cap: usize,
                            proof_len: *mut usize) -> c_int;
    pub fn bpg_prover_prove_many(provers: *const *mut bpg_prover, k: usize, rng: bpg_rng_fn, rng_user: *mut c_void,
                                 proofs_out: *mut u8, cap_each: usize, proof_lens: *mut usize) -> c_int;
    pub fn bpg_prover_begin_blindings(p: *mut bpg_prover, ext32: *const u8) -> c_int;
    pub fn bpg_prover_stats(p: *const bpg_prover, multipliers: *mut usize, constraints: *mut usize) -> c_int;
    pub fn bpg_prover_timing(p: *const bpg_prover, out_ms: *mut c_double) -> c_int;
    pub fn bpg_prover_kernel_timing(p: *const bpg_prover, out_us: *mut c_double, out_bytes: *mut c_double, out_montmuls: *mut c_double) -> c_int;
    // ---- verifier
    pub fn bpg_verifier_new(label: *const u8, label_len: usize, out: *mut *mut bpg_verifier) -> c_int;
    pub fn bpg_verifier_free(v: *mut bpg_verifier);
    pub fn bpg_verifier_cs(v: *mut bpg_verifier) -> *mut bpg_cs;
    pub fn bpg_verifier_commit(v: *mut bpg_verifier, V32: *const u8, var: *mut u64) -> c_int;
    pub fn bpg_verifier_commit_batch(v: *mut bpg_verifier, V32: *const u8, m: usize, vars: *mut u64) -> c_int;
    pub fn bpg_verifier_verify(v: *mut bpg_verifier, gens: *mut bpg_gens, proof: *const u8, proof_len: usize, rng: bpg_rng_fn,
                               rng_user: *mut c_void) -> c_int;
    pub fn bpg_verifier_verify_batch(verifiers: *const *mut bpg_verifier, proofs: *const *const u8, proof_lens: *const usize,
                                     k: usize, gens: *mut bpg_gens, rng: bpg_rng_fn, rng_user: *mut c_void) -> c_int;
    pub fn bpg_verifier_stats(v: *const bpg_verifier, multipliers: *mut usize, constraints: *mut usize) -> c_int;
    // ---- constraint system
    pub fn bpg_cs_multiply(cs: *mut bpg_cs, lvars: *const u64, lcoeffs: *const u8, nl: usize, rvars: *const u64,
                           rcoeffs: *const u8, nr: usize, out: *mut u64) -> c_int;
    pub fn bpg_cs_allocate_multiplier(cs: *mut bpg_cs, left32: *const u8, right32: *const u8, out: *mut u64) -> c_int;
    pub fn bpg_cs_constrain(cs: *mut bpg_cs, vars: *const u64, coeffs: *const u8, n: usize) -> c_int;
    pub fn bpg_cs_specify_randomized_constraints(cs: *mut bpg_cs, callback: bpg_randomized_fn, user: *mut c_void) -> c_int;
    pub fn bpg_cs_challenge_scalar(rcs: *mut bpg_cs, label: *const u8, label_len: usize, out32: *mut u8) -> c_int;
    pub fn bpg_flat_cache_stats(ctx: *mut bpg_ctx, hits: *mut u64, misses: *mut u64) -> c_int;
    pub fn bpg_cs_export_terms(cs: *const bpg_cs, vars: *mut u64, coeffs32: *mut u8, constraint: *mut u32, cap: usize,
                               n_terms: *mut usize) -> c_int;
    // ---- the reference's gadget library
    pub fn bpg_gadget_binary_constrain(cs: *mut bpg_cs, bit_var: u64) -> c_int;
    pub fn bpg_gadget_nonzero(cs: *mut bpg_cs, vars: *const u64, coeffs: *const u8, n: usize, assignment32: *const u8) -> c_int;
    pub fn bpg_point_from_coords(xyzt128: *const u8, out: *mut *mut bpg_point_gadget) -> c_int;
    pub fn bpg_point_from_vars(vars: *const u64, out: *mut *mut bpg_point_gadget) -> c_int;
    pub fn bpg_point_free(p: *mut bpg_point_gadget);
    pub fn bpg_edwards_add(cs: *mut bpg_cs, a: *const bpg_point_gadget, b: *const bpg_point_gadget, out: *mut *mut bpg_point_gadget) -> c_int;
    pub fn bpg_edwards_double(cs: *mut bpg_cs, a: *const bpg_point_gadget, out: *mut *mut bpg_point_gadget) -> c_int;
    pub fn bpg_edwards_equal(cs: *mut bpg_cs, a: *const bpg_point_gadget, b: *const bpg_point_gadget) -> c_int;
    pub fn bpg_edwards_satisfy_curve_eq(cs: *mut bpg_cs, a: *const bpg_point_gadget) -> c_int;
    pub fn bpg_edwards_conditionally_select(cs: *mut bpg_cs, a: *const bpg_point_gadget, bit_vars: *const u64, bit_coeffs: *const u8,
                                            n: usize, out: *mut *mut bpg_point_gadget) -> c_int;
    pub fn bpg_edwards_scalar_mul(cs: *mut bpg_cs, point: *const bpg_point_gadget, sk_vars: *const u64, nbits: usize,
                                  out: *mut *mut bpg_point_gadget) -> c_int;
    pub fn bpg_edwards_prover_commit(p: *mut bpg_prover, xyzt128: *const u8, blindings128: *const u8, out: *mut *mut bpg_point_gadget,
                                     V128: *mut u8) -> c_int;
    pub fn bpg_edwards_verifier_commit(v: *mut bpg_verifier, V32: *const u8, n_commitments: usize, out: *mut *mut bpg_point_gadget) -> c_int;
    pub fn bpg_ristretto_from_point(cs: *mut bpg_cs, xyzt128: *const u8, out: *mut *mut bpg_point_gadget) -> c_int;
    pub fn bpg_ristretto_from_lcs(cs: *mut bpg_cs, lcs: *const bpg_point_gadget, n_lcs: usize, out: *mut *mut bpg_point_gadget) -> c_int;
    pub fn bpg_ristretto_add(cs: *mut bpg_cs, a: *const bpg_point_gadget, b: *const bpg_point_gadget, out: *mut *mut bpg_point_gadget) -> c_int;
    pub fn bpg_ristretto_double(cs: *mut bpg_cs, a: *const bpg_point_gadget, out: *mut *mut bpg_point_gadget) -> c_int;
    pub fn bpg_ristretto_equals(cs: *mut bpg_cs, a: *const bpg_point_gadget, b: *const bpg_point_gadget) -> c_int;
    pub fn bpg_ristretto_conditionally_select(cs: *mut bpg_cs, a: *const bpg_point_gadget, bit_vars: *const u64, bit_coeffs: *const u8,
                                              n: usize, out: *mut *mut bpg_point_gadget) -> c_int;
    pub fn bpg_sk_knowledge_gadget(cs: *mut bpg_cs, basep: *const bpg_point_gadget, pk: *const bpg_point_gadget, sk_vars: *const u64,
                                   nbits: usize) -> c_int;
    // ---- Merlin known-answer hook
    pub fn bpg_transcript_kat(label: *const u8, label_len: usize, msg_label: *const c_char, msg: *const u8, msg_len: usize,
                              challenge_label: *const c_char, out: *mut u8, out_len: usize) -> c_int;
}
