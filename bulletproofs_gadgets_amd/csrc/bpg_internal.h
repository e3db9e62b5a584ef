// bpg_internal.h -- shared declarations of the MI355X product library (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/bpg.h"
#include "ge25519.h"
#include "sc25519.h"

namespace bpg {

// One resident table entry: affine Niels point, one 128-B line.
struct alignas(16) aniels_rec {
    uint32_t w[32];   // [0..8) (y+x)/2, [8..16) (y-x)/2, [16..24) dxy, [24..32) -dxy (the entry of -P is the same record re-read)
};
static_assert(sizeof(aniels_rec) == 128, "table stride is one cache line");

// Extended point as stored in HBM work buffers (128 B): X, Y, Z, T.
struct alignas(16) ge_rec {
    uint32_t w[32];
};

void set_last_error(const char *what, hipError_t e);
void set_last_error_msg(const char *what);

#define BPG_CHECK_HIP(expr)                                   \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) {                               \
            bpg::set_last_error(#expr, _e);                   \
            return BPG_EDEVICE;                               \
        }                                                     \
    } while (0)

// Geometry of one Pippenger run.
struct msm_plan {
    int c;            // window bits
    int W;            // digit windows = ceil(252 / c)
    int Wb;           // bucket sets per target: W (variable-base) or 1 (fixed-base: all windows share one)
    int NB;           // buckets per window = 2^(c-1)
    int P;            // entries per accumulate thread (= ls0)
    int ls0, ls1;     // ls0: S0, entries per accumulate thread (any integer); ls1: log2 of S1, partials per merge thread
    int R;            // reduction passes after the accumulate pass
    int tile;         // scalars per histogram tile
    int ntiles;
    size_t n;         // scalars per target
    int T;            // targets
    int chunk;        // buckets per tail-A block = min(256, NB)
    int nchunks;      // NB / chunk
    int outs;         // points per (target, window) = c  (total + c-1 bit sums)
    int nbin;         // bins per bucket set in the two-level sort (NB = nbin << lowshift)
    int lowshift;     // log2(buckets per bin)
    int sort_chunk;   // items of one bin that one block of the level-2 sort handles
};
struct msm_tuning {
    int c = 0, tile = 0, ls0 = 0, ls1 = 0;   // 0 = automatic (BPG_MSM_C / _TILE / _LS0 / _LS1 env overrides)
    int sort_chunk = 0;                      // 0 = automatic (BPG_MSM_SORT_CHUNK)
    int ipp_fold = -1;                       // inner-product round after which the folded generators are materialized
                                             // (BPG_IPP_FOLD; 0 = never, -1 = default)
};
msm_plan make_plan(size_t n, int T, const msm_tuning &tune, int fixed_c = 0, float density = 1.f, int resident_lanes = 262144);
// fixed-base (precomputed 2^(c w) multiples) geometry for a table of `points` generators
int fixed_base_window(size_t points);
int fixed_base_windows(int c);

int grow(void **p, size_t *cap, size_t need);
// pinned host memory, same contract as grow() (BPG_OK or BPG_ENOMEM; the old contents are not kept)
int grow_pinned(void **p, size_t *cap, size_t need);
// hipMalloc behind the fault-injection hook of bpg_debug_fail_alloc (every growable workspace goes through it)
hipError_t device_malloc(void **p, size_t bytes);
void debug_fail_alloc(int countdown);

inline void ge_load(ge &p, const ge_rec &r) {
    for (int i = 0; i < 8; i++) {
        p.X.v[i] = r.w[i];
        p.Y.v[i] = r.w[8 + i];
        p.Z.v[i] = r.w[16 + i];
        p.T.v[i] = r.w[24 + i];
    }
}

}  // namespace bpg

struct bpg_points {
    bpg_ctx *ctx;
    bpg::aniels_rec *d_table;
    size_t n;
};

struct bpg_lane;

// Device workspace + stream of one in-flight MSM.
struct bpg_msm_job {
    bpg_lane *lane = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // start, acc begin, acc end, end
    bpg::msm_plan plan{};
    int16_t *d_digits = nullptr;      size_t cap_digits = 0;
    uint32_t *d_entries = nullptr;    size_t cap_entries = 0;
    uint32_t *d_counts = nullptr;     size_t cap_counts = 0;
    bpg::ge_rec *d_partials = nullptr; size_t cap_partials = 0;
    bpg::ge_rec *d_chunk = nullptr;   size_t cap_chunk = 0;
    bpg::ge_rec *d_out = nullptr;     size_t cap_out = 0;
    bpg::ge_rec *h_out = nullptr;     size_t cap_hout = 0;      // pinned host
    void *d_stage = nullptr;          size_t cap_stage = 0;     // scalars staged from the host
    void *d_tmp = nullptr;            size_t cap_tmp = 0;       // bin-partitioned entries (level 1 of the sort)
    std::atomic<bool> busy{false};
};

// Everything mutable that a call needs on the device.  One lane per host thread that
// has called into a context (created on that thread's first call), so N host threads
// drive N proofs on N streams against shared, immutable generator / point tables
// ("one proof = one thread").
struct bpg_lane {
    bpg_ctx *ctx = nullptr;
    hipStream_t stream = nullptr;     // setup / codec / scalar-vector kernel stream
    hipEvent_t ev_k[12] = {};         // start / stop events around the six scalar-vector kernels of a single proof (ctx->time_kernels)
    hipEvent_t ev_sync = nullptr;     // waits on `stream` go through this event (lane_sync); null when the context spins
    std::vector<bpg_msm_job *> jobs;  // MSM job pool (each job has its own stream)
    float last_acc_us = 0.f, last_total_us = 0.f;
    bpg_msm_job *acc_gate_job = nullptr;   // job whose accumulate pass was submitted last (its ev[2] gates the next one)
    // bump arena for the R1CS scalar-vector kernels (grown on demand, reused across proofs)
    void *ws = nullptr;
    size_t ws_cap = 0;
    // per-proof points of a verification (commitments, proof points), decoded to affine Niels
    bpg::aniels_rec *d_dyn = nullptr;
    size_t dyn_cap = 0;
    // staging of the small codec / Pedersen kernels (inputs and outputs of one call; grown on demand, never freed
    // per call: hipFree is a device-wide synchronisation, and 24 prover threads share the GPU)
    void *d_io = nullptr;
    size_t io_cap = 0;
    // batch verification: the accumulated generator scalars (2 x capacity) and the per-proof point scalars of the
    // combined equation.  They outlive the per-proof passes over the arena above, so they have their own buffers.
    // pinned host staging (hipHostMalloc; grown on demand, kept): the blinding draws are written here as the RNG
    // delivers them and the witness vectors are gathered here, so every large upload of a proof is a plain DMA.
    // (hipMemcpyAsync from pageable memory goes through the runtime's own staging path: under 24 prover threads
    // it cost 5 ms of CPU per 2^16 proof, most of it waiting inside the runtime.)
    void *h_raw = nullptr;
    size_t h_raw_cap = 0;
    void *h_wit = nullptr;
    size_t h_wit_cap = 0;
    // ... and every small transfer of a proof (challenge tables, blindings, partial sums, commitments) is staged here:
    // a copy between pageable memory and the device makes the runtime wait inside the call, spinning unless the
    // device's blocking-sync flag is set -- and that flag is not safe to set (api.hip: bpg_init)
    void *h_io = nullptr;             // prove / verify (r1cs.hip: PinArena)
    size_t h_io_cap = 0;
    void *h_ped = nullptr;            // Pedersen batches
    size_t h_ped_cap = 0;
    void *h_dec = nullptr;            // compressed points on their way to k_decode
    size_t h_dec_cap = 0;
    void *d_vacc = nullptr;
    size_t vacc_cap = 0;
    void *d_vtail = nullptr;
    size_t vtail_cap = 0;
};

struct bpg_ctx {
    int device = 0;
    uint64_t id = 0;                  // process-unique; keys the per-thread lane cache
    bpg::msm_tuning tune;
    int resident_lanes = 262144;      // CUs x 4 SIMDs x 4 waves x 64 lanes: threads of k_accumulate resident at once
    bool time_kernels = false;        // BPG_TIME_KERNELS=1: a single proof times its scalar-vector kernels (bpg_prover_kernel_timing)
    bool sync_block = true;           // waits sleep instead of spinning; BPG_SYNC=spin clears it
    bool sync_poll = false;           // waits always poll and nap (BPG_SYNC=poll, or the device's blocking-sync flag could not be set)
    std::mutex mu;                    // guards lanes, free_lanes
    std::vector<bpg_lane *> lanes;
    std::vector<bpg_lane *> free_lanes;   // lanes of threads that have exited: the next new thread takes one over
    // this device's copy of the static workspace of bpg_msm_noalloc (msm.hip: msm_noalloc_init)
    void *noalloc_scratch = nullptr;
    size_t noalloc_scratch_bytes = 0;
    // circuit-structure cache of the constraint flattening (r1cs.hip: flat_plan_cached): the counting-sorted term
    // lists of recently seen circuits, resident in HBM; most recently used last.  Entries are immutable once listed.
    std::mutex flat_mu;
    std::vector<std::shared_ptr<void>> flat_cache;
    std::atomic<uint64_t> flat_hits{0}, flat_misses{0};
    std::atomic<uint64_t> flat_followed{0};          // hits found by following the cached circuit during capture (no hash, no compare)
    // pinned buffers of early blinding draws (r1cs.hip: Prover::begin_blindings), kept for the next large proof
    std::mutex spec_mu;
    std::vector<std::pair<void *, size_t>> spec_pool;
};

namespace bpg {
// Host-phase accounting (BPG_TRACE_HOST=1; printed by bpg_shutdown): wall and thread-CPU time per named phase,
// summed over all threads.  Where a prover thread's time goes under load cannot be read off a GPU trace.
struct HostPhase {
    enum Id { CAPTURE_COMMIT, RNG_BEGIN, RNG_DRAW, UPLOAD, AS_SUBMIT, AS_WAIT, FLATTEN_POLY, T_COMMIT, IPP_ROWS, IPP_SUBMIT, IPP_WAIT, IPP_HOST,
              FOLD, TAIL_HOST, GADGET, N };
    static bool enabled();
    static void add(Id id, double wall_ms, double cpu_ms);
    static void report();
    Id id;
    double w0, c0;
    bool on;
    explicit HostPhase(Id i);
    void stop();                       // account now (the destructor then does nothing)
    ~HostPhase() { stop(); }
};
// the calling thread's lane of ctx (created on first use, device made current); nullptr on HIP failure
bpg_lane *lane_of(bpg_ctx *ctx);
// wait for the lane's stream (sleeping or spinning, per ctx->sync_block)
hipError_t lane_sync(bpg_lane *lane);
hipError_t wait_event(const bpg_ctx *ctx, hipEvent_t ev);
// (api.hip) dst[0..n16) = src[0..n16) in 16-byte units, by the device
__global__ void k_copy16(uint4 *__restrict__ dst, const uint4 *__restrict__ src, uint32_t n16);

// One run of consecutive table records that an MSM reads; up to 4 per MSM, possibly from
// different tables (generator table + a lane's per-proof points).
struct msm_seg {
    const aniels_rec *base;
    size_t len;
    size_t sc_off = 0;     // first scalar of the segment within a target's row (only read when sc_stride != 0)
};
// Fixed-base mode: fb.c != 0.  The segments then point into window 0 of a shifted table in
// which record (w * fb.wstride + i) holds 2^(c w) P_i, and all windows share one bucket set.
struct msm_fixed_base {
    int c = 0;
    size_t wstride = 0;
    float density = 1.f;    // hint: expected fraction of non-zero scalars (sizes the accumulate segments)
};
int msm_submit(bpg_ctx *ctx, const msm_seg *segs, int nseg, const void *scalars, size_t n, int T, bool scalars_on_host,
               bpg_msm_job **out_job, msm_fixed_base fb = msm_fixed_base(), size_t sc_stride = 0);
// sc_stride = 0: target t's scalars are the dense row scalars[t*n .. t*n+n), in segment order.
// sc_stride != 0 (device scalars only): rows are sc_stride scalars apart and segment k reads
// scalars[t*sc_stride + segs[k].sc_off ..): a rank of a sharded MSM reads its slices in place.
int table_write_uniform(bpg_ctx *ctx, aniels_rec *dst, const uint8_t *host64, size_t n);
// pending_bad != nullptr: where possible (few points) nothing is waited for -- *pending_bad then points at the count of
// invalid encodings, valid after the caller's next wait on the lane's stream; nullptr in *pending_bad: *n_bad is set
int table_write_compressed(bpg_ctx *ctx, aniels_rec *dst, const uint8_t *host32, size_t n, uint32_t *n_bad,
                           const uint32_t **pending_bad = nullptr);
// table holds W * npts records with records [0, npts) filled: fills [w * npts + i] = 2^(c w) P_i
int table_build_shifted(bpg_ctx *ctx, aniels_rec *table, size_t npts, int c, int W);
int msm_wait(bpg_msm_job *job, uint8_t *out, bool partial);
int msm_noalloc_init(bpg_ctx *ctx);
int msm_noalloc(bpg_ctx *ctx, const msm_seg *seg, const void *d_scalars, size_t n, void *d_out32, hipStream_t s);
int msm_fold_submit(bpg_ctx *ctx, const msm_seg segs[2], const void *d_scalars, size_t stride, int rows, size_t np, size_t m,
                    msm_fixed_base fb, aniels_rec *d_tab, bpg_msm_job **out_job);
int msm_fold_wait(bpg_msm_job *job);
int combine_partials(const uint8_t *ext128, size_t k, uint8_t out32[32]);
void msm_job_destroy(bpg_msm_job *job);
void forget_shapes_of(uint64_t ctx_id);
}  // namespace bpg
