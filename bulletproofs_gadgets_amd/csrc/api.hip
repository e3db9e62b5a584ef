// api.hip -- C-ABI entry points (include/bpg.h): context, device memory helpers,
// resident point tables (batch ristretto decode / Elligator / compress kernels),
// and the MSM entry points.  No CPU fallback: bpg_init fails without a GPU.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <sys/prctl.h>

#include "bpg_internal.h"

namespace bpg {

static thread_local char g_err[512] = "";

void set_last_error(const char *what, hipError_t e) {
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
}
void set_last_error_msg(const char *what) { snprintf(g_err, sizeof g_err, "%s", what); }

__device__ __forceinline__ void store_aniels(aniels_rec *rec, const ge_aniels &q) {
    uint4 *p = reinterpret_cast<uint4 *>(rec);
    p[0] = make_uint4(q.hYpX.v[0], q.hYpX.v[1], q.hYpX.v[2], q.hYpX.v[3]);
    p[1] = make_uint4(q.hYpX.v[4], q.hYpX.v[5], q.hYpX.v[6], q.hYpX.v[7]);
    p[2] = make_uint4(q.hYmX.v[0], q.hYmX.v[1], q.hYmX.v[2], q.hYmX.v[3]);
    p[3] = make_uint4(q.hYmX.v[4], q.hYmX.v[5], q.hYmX.v[6], q.hYmX.v[7]);
    p[4] = make_uint4(q.Tdxy.v[0], q.Tdxy.v[1], q.Tdxy.v[2], q.Tdxy.v[3]);
    p[5] = make_uint4(q.Tdxy.v[4], q.Tdxy.v[5], q.Tdxy.v[6], q.Tdxy.v[7]);
    // fourth slot: -2dxy, so that the record of -P is this record read in another order (msm.hip: load_aniels_signed)
    fe nt;
    fe_neg(nt, q.Tdxy);
    p[6] = make_uint4(nt.v[0], nt.v[1], nt.v[2], nt.v[3]);
    p[7] = make_uint4(nt.v[4], nt.v[5], nt.v[6], nt.v[7]);
}

// one contiguous move of 16-byte units by the device itself (pinned host memory on one side): a kernel on the stream
// instead of a copy-engine command (r1cs.hip: PinArena)
__global__ void __launch_bounds__(256) k_copy16(uint4 *__restrict__ dst, const uint4 *__restrict__ src, uint32_t n16) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) dst[i] = src[i];
}

// CompressedRistretto::decompress per thread; invalid encodings raise *bad and store the identity
__global__ void __launch_bounds__(256) k_decode(const uint8_t *__restrict__ in32, aniels_rec *__restrict__ table, size_t n,
                                                uint32_t *__restrict__ bad) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint4 *p = reinterpret_cast<const uint4 *>(in32 + 32 * i);
    uint4 a0 = p[0], a1 = p[1];
    uint32_t b[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    fe x, y;
    ge_aniels q;
    if (ristretto_decode_affine_w(x, y, b)) {
        ge_affine_to_aniels(q, x, y);
    } else {
        ge_aniels_identity(q);
        atomicAdd(bad, 1u);
    }
    store_aniels(table + i, q);
}

// RistrettoPoint::from_uniform_bytes per thread, normalised to affine Niels
__global__ void __launch_bounds__(256) k_from_uniform(const uint8_t *__restrict__ in64, aniels_rec *__restrict__ table,
                                                      size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint4 *p = reinterpret_cast<const uint4 *>(in64 + 64 * i);
    uint4 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3];
    uint32_t b[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
    ge P;
    ristretto_from_uniform_words(P, b);
    fe zinv;
    fe_invert(zinv, P.Z);
    ge_aniels q;
    ge_to_aniels(q, P, zinv);
    store_aniels(table + i, q);
}

// RistrettoPoint::compress of table entries
__global__ void __launch_bounds__(256) k_compress(const aniels_rec *__restrict__ table, uint8_t *__restrict__ out32, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t *w = table[i].w;
    ge_aniels q;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        q.hYpX.v[k] = w[k];
        q.hYmX.v[k] = w[8 + k];
    }
    ge P;
    ge_from_aniels(P, q);
    uint32_t b[8];
    ristretto_encode_w(b, P);
    uint4 *o = reinterpret_cast<uint4 *>(out32 + 32 * i);
    o[0] = make_uint4(b[0], b[1], b[2], b[3]);
    o[1] = make_uint4(b[4], b[5], b[6], b[7]);
}

static std::atomic<uint64_t> g_next_ctx_id{1};

// live contexts by id: a thread that exits hands its lanes back to the contexts that still exist
static std::mutex g_live_mu;
static std::vector<bpg_ctx *> g_live;

namespace {
struct LaneSlot {
    uint64_t id;
    bpg_lane *lane;
};
// A lane holds a thread's streams and workspaces (gigabytes for grouped proofs): threads come and go (a pool that
// is rebuilt, one thread per request), so a lane outlives its thread and is taken over by the next new one.
struct LaneCache {
    std::vector<LaneSlot> slots;
    ~LaneCache() {
        std::lock_guard<std::mutex> lk(g_live_mu);
        for (auto &s : slots)
            for (bpg_ctx *c : g_live)
                if (c->id == s.id) {
                    std::lock_guard<std::mutex> lk2(c->mu);
                    c->free_lanes.push_back(s.lane);
                }
    }
};
}  // namespace

bpg_lane *lane_of(bpg_ctx *ctx) {
    static thread_local LaneCache tl;
    auto &cache = tl.slots;
    if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
    for (auto &s : cache)
        if (s.id == ctx->id) return s.lane;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        if (!ctx->free_lanes.empty()) {
            bpg_lane *lane = ctx->free_lanes.back();
            ctx->free_lanes.pop_back();
            cache.push_back({ctx->id, lane});
            return lane;
        }
    }
    bpg_lane *lane = new bpg_lane();
    lane->ctx = ctx;
    hipError_t e = hipStreamCreateWithFlags(&lane->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_last_error("hipStreamCreateWithFlags(lane)", e);
        delete lane;
        return nullptr;
    }
    if (ctx->sync_block) {
        e = hipEventCreateWithFlags(&lane->ev_sync, hipEventDisableTiming);
        if (e != hipSuccess) {
            set_last_error("hipEventCreateWithFlags(lane)", e);
            (void)hipStreamDestroy(lane->stream);
            delete lane;
            return nullptr;
        }
    }
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        ctx->lanes.push_back(lane);
    }
    cache.push_back({ctx->id, lane});
    return lane;
}

// ---- host-phase accounting (BPG_TRACE_HOST=1)
static std::atomic<int> g_trace_host{-1};
static std::atomic<uint64_t> g_phase_wall_us[HostPhase::N], g_phase_cpu_us[HostPhase::N], g_phase_calls[HostPhase::N];
static double clock_ms(clockid_t id) {
    struct timespec ts;
    clock_gettime(id, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
bool HostPhase::enabled() {
    int v = g_trace_host.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("BPG_TRACE_HOST");
        v = e && *e && *e != '0';
        g_trace_host.store(v);
    }
    return v != 0;
}
void HostPhase::add(Id id, double wall_ms, double cpu_ms) {
    g_phase_wall_us[id] += (uint64_t)(wall_ms * 1e3);
    g_phase_cpu_us[id] += (uint64_t)(cpu_ms * 1e3);
    g_phase_calls[id]++;
}
HostPhase::HostPhase(Id i) : id(i), w0(0), c0(0), on(enabled()) {
    if (on) {
        w0 = clock_ms(CLOCK_MONOTONIC);
        c0 = clock_ms(CLOCK_THREAD_CPUTIME_ID);
    }
}
void HostPhase::stop() {
    if (on) add(id, clock_ms(CLOCK_MONOTONIC) - w0, clock_ms(CLOCK_THREAD_CPUTIME_ID) - c0);
    on = false;
}
void HostPhase::report() {
    if (!enabled()) return;
    static const char *names[N] = {"capture_commit", "rng_begin", "rng_draw", "upload", "as_submit", "as_wait", "flatten_poly", "t_commit",
                                   "ipp_rows", "ipp_submit", "ipp_wait", "ipp_host", "fold", "tail_host", "gadget_calls"};
    fprintf(stderr, "[bpg host phases] %-16s %10s %12s %12s\n", "phase", "calls", "wall_ms", "thread_cpu_ms");
    for (int i = 0; i < N; i++)
        fprintf(stderr, "[bpg host phases] %-16s %10llu %12.1f %12.1f\n", names[i], (unsigned long long)g_phase_calls[i].load(),
                g_phase_wall_us[i].load() / 1e3, g_phase_cpu_us[i].load() / 1e3);
}

// Waiting without burning a core, measured at 2^16-multiplier proofs:
//  * spinning (the runtime's default; BPG_SYNC=spin): 36.6 ms of thread CPU per proof at 16 threads.  Lowest latency for
//    one proof alone, but a prover service runs more threads than cores and a spinning waiter takes a core from a
//    thread that captures or draws.
//  * polling the event with naps of 20-100 us (the default; BPG_SYNC=poll): 0.7 ms of CPU per proof inside the waits at
//    48 threads.  Every transfer of the hot paths goes through pinned staging (PinArena, h_raw, h_wit), so no wait
//    hides inside the runtime where it would spin.
//  * the device's scheduling flag hipDeviceScheduleBlockingSync (BPG_SYNC=block) makes the runtime sleep on an
//    interrupt: 0.9 ms of CPU per proof inside the waits at 24 threads, 10.4 ms at 48 (every completed kernel wakes
//    every sleeper), and 5-10 % more proofs/s than polling because a nap overshoots.  NOT the default: on ROCm 7.2 / this
//    driver, once a process with the flag set has run a few thousand proofs from 16 threads, the first HIP call of any
//    process started afterwards (hipGetDeviceCount in a child, while the parent lives) takes SIGSEGV inside the
//    runtime's initialisation -- reproduced with the flag set and never without it.  A service that
//    never starts another GPU process beside the prover may opt in.  (hipEventBlockingSync on an event alone still
//    spins in user mode; a hybrid of interrupt and polling was within noise of the plain interrupt.)
hipError_t wait_event(const bpg_ctx *ctx, hipEvent_t ev) {
    if (!ctx->sync_block || !ctx->sync_poll) return hipEventSynchronize(ev);
    hipError_t e;
    for (int spin = 0; spin < 20; spin++)
        if ((e = hipEventQuery(ev)) != hipErrorNotReady) return e;
    static const long nap0 = getenv("BPG_NAP0") ? atol(getenv("BPG_NAP0")) : 20000, nap1 = getenv("BPG_NAP1") ? atol(getenv("BPG_NAP1")) : 100000;
    static const long slack = getenv("BPG_SLACK") ? atol(getenv("BPG_SLACK")) : 0;
    static thread_local bool slack_set = false;
    if (slack && !slack_set) {
        slack_set = true;
        prctl(PR_SET_TIMERSLACK, (unsigned long)slack, 0, 0, 0);
    }
    long ns = nap0;
    while ((e = hipEventQuery(ev)) == hipErrorNotReady) {
        struct timespec ts = {0, ns};
        nanosleep(&ts, nullptr);
        if (ns < nap1) ns += 10000;
    }
    return e;
}
hipError_t lane_sync(bpg_lane *lane) {
    const bpg_ctx *ctx = lane->ctx;
    if (!ctx->sync_block || !lane->ev_sync) return hipStreamSynchronize(lane->stream);
    hipError_t e = hipEventRecord(lane->ev_sync, lane->stream);
    if (e != hipSuccess) return e;
    return wait_event(ctx, lane->ev_sync);
}

#define BPG_LANE(ctx)                    \
    bpg_lane *lane = bpg::lane_of(ctx);  \
    if (!lane) return BPG_EDEVICE

// Fill table entries [dst, dst + n) from host bytes (used by the generator table and by
// the verifier's per-proof points).  uniform: 64 B each (from_uniform_bytes);
// compressed: 32 B each (decompress); *n_bad receives the number of invalid encodings.
int table_write_uniform(bpg_ctx *ctx, aniels_rec *dst, const uint8_t *host64, size_t n) {
    if (!n) return BPG_OK;
    BPG_LANE(ctx);
    int rc;
    if ((rc = grow(&lane->d_io, &lane->io_cap, n * 64))) return rc;
    uint8_t *d_in = (uint8_t *)lane->d_io;
    BPG_CHECK_HIP(hipMemcpyAsync(d_in, host64, n * 64, hipMemcpyHostToDevice, lane->stream));
    hipLaunchKernelGGL(k_from_uniform, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, lane->stream, d_in, dst, n);
    BPG_CHECK_HIP(hipGetLastError());
    BPG_CHECK_HIP(lane_sync(lane));
    return BPG_OK;
}

int table_write_compressed(bpg_ctx *ctx, aniels_rec *dst, const uint8_t *host32, size_t n, uint32_t *n_bad, const uint32_t **pending_bad) {
    *n_bad = 0;
    if (!n) return BPG_OK;
    BPG_LANE(ctx);
    int rc;
    if ((rc = grow(&lane->d_io, &lane->io_cap, n * 32 + 256))) return rc;
    uint8_t *d_in = (uint8_t *)lane->d_io;
    uint32_t *d_flag = (uint32_t *)(d_in + n * 32);
    // a proof's few dozen points (the verifier's commitments, T's, L's and R's): staged in pinned memory of the lane's
    // own, read there by k_decode, and the count of bad encodings comes back through a 16-byte kernel copy -- no
    // copy-engine command, no pageable transfer (the runtime waits inside those, spinning)
    if (n <= 8192 && grow_pinned(&lane->h_dec, &lane->h_dec_cap, n * 32 + 64) == BPG_OK) {
        uint8_t *h_in = (uint8_t *)lane->h_dec;
        uint32_t *h_flag = (uint32_t *)(h_in + n * 32 + ((64 - (n * 32) % 64) % 64));
        memcpy(h_in, host32, n * 32);
        BPG_CHECK_HIP(hipMemsetAsync(d_flag, 0, 16, lane->stream));
        hipLaunchKernelGGL(k_decode, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, lane->stream, (const uint8_t *)h_in, dst, n, d_flag);
        hipLaunchKernelGGL(k_copy16, dim3(1), dim3(256), 0, lane->stream, (uint4 *)h_flag, (const uint4 *)d_flag, 1u);
        BPG_CHECK_HIP(hipGetLastError());
        if (pending_bad) {                     // the caller has more to enqueue before it waits: the count is *pending_bad then
            *pending_bad = h_flag;
            return BPG_OK;
        }
        BPG_CHECK_HIP(lane_sync(lane));
        *n_bad = h_flag[0];
        return BPG_OK;
    }
    if (pending_bad) *pending_bad = nullptr;   // (not taken: the count is in *n_bad on return)
    BPG_CHECK_HIP(hipMemcpyAsync(d_in, host32, n * 32, hipMemcpyHostToDevice, lane->stream));
    BPG_CHECK_HIP(hipMemsetAsync(d_flag, 0, 4, lane->stream));
    hipLaunchKernelGGL(k_decode, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, lane->stream, d_in, dst, n, d_flag);
    BPG_CHECK_HIP(hipGetLastError());
    BPG_CHECK_HIP(hipMemcpyAsync(n_bad, d_flag, 4, hipMemcpyDeviceToHost, lane->stream));
    BPG_CHECK_HIP(lane_sync(lane));
    return BPG_OK;
}

// Fixed-base table: records [w * npts + i] = 2^(c w) P_i for w = 1 .. W-1, from window 0.
// One thread per point walks the doubling chain, parks X | Y | Z | (running product of the Z's)
// in the 128-byte record, then converts its W-1 points to affine Niels with ONE inversion.
__global__ void __launch_bounds__(256) k_shift_table(aniels_rec *__restrict__ table, size_t npts, int c, int W) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npts) return;
    ge p;
    {
        const uint4 *q = reinterpret_cast<const uint4 *>(table + i);
        uint4 a0 = q[0], a1 = q[1], b0 = q[2], b1 = q[3];
        ge_aniels q0;
        q0.hYpX = {{a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}};
        q0.hYmX = {{b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w}};
        ge_from_aniels(p, q0);
    }
    fe prod;
    fe_set1(prod);
    for (int w = 1; w < W; w++) {
        for (int k = 0; k < c; k++) ge_double_xyz(p, p);      // T is never read: the chain only doubles, the records take X, Y, Z
        fe_mul(prod, prod, p.Z);
        ge tmp = p;
        tmp.T = prod;
        uint4 *o = reinterpret_cast<uint4 *>(table + (size_t)w * npts + i);
        const fe *f[4] = {&tmp.X, &tmp.Y, &tmp.Z, &tmp.T};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            o[2 * k] = make_uint4(f[k]->v[0], f[k]->v[1], f[k]->v[2], f[k]->v[3]);
            o[2 * k + 1] = make_uint4(f[k]->v[4], f[k]->v[5], f[k]->v[6], f[k]->v[7]);
        }
    }
    fe inv;
    fe_invert(inv, prod);                  // 1 / (Z_1 ... Z_{W-1})
    for (int w = W - 1; w >= 1; w--) {
        aniels_rec *rec = table + (size_t)w * npts + i;
        const uint4 *q = reinterpret_cast<const uint4 *>(rec);
        fe X, Y, Z, prev;
        uint4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3], v4 = q[4], v5 = q[5];
        X = {{v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w}};
        Y = {{v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w}};
        Z = {{v4.x, v4.y, v4.z, v4.w, v5.x, v5.y, v5.z, v5.w}};
        if (w > 1) {
            const uint4 *qp = reinterpret_cast<const uint4 *>(table + (size_t)(w - 1) * npts + i);
            uint4 t0 = qp[6], t1 = qp[7];
            prev = {{t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w}};
        } else {
            fe_set1(prev);
        }
        fe zinv, x, y;
        fe_mul(zinv, inv, prev);           // 1 / Z_w
        fe_mul(inv, inv, Z);               // 1 / (Z_1 ... Z_{w-1})
        fe_mul(x, X, zinv);
        fe_mul(y, Y, zinv);
        ge_aniels n;
        ge_affine_to_aniels(n, x, y);
        store_aniels(rec, n);
    }
}

int table_build_shifted(bpg_ctx *ctx, aniels_rec *table, size_t npts, int c, int W) {
    if (!npts || W < 2) return BPG_OK;
    BPG_LANE(ctx);
    hipLaunchKernelGGL(k_shift_table, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, lane->stream, table, npts, c, W);
    BPG_CHECK_HIP(hipGetLastError());
    BPG_CHECK_HIP(lane_sync(lane));
    return BPG_OK;
}

}  // namespace bpg

using namespace bpg;

extern "C" {

const char *bpg_last_error(void) { return bpg::g_err; }
const char *bpg_version(void) { return "bulletproofs_gadgets_amd 0.1 (gfx950)"; }

int bpg_init(int device_id, bpg_ctx **out) {
    if (!out) return BPG_EINVAL;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_last_error_msg("bpg_init: no HIP device visible (this library has no CPU fallback)");
        return BPG_EDEVICE;
    }
    if (device_id < 0 || device_id >= count) return BPG_EINVAL;
    {
        // the kernels are built for gfx950 (MI355X) only: anything else would fail at the first launch
        hipDeviceProp_t prop;
        BPG_CHECK_HIP(hipGetDeviceProperties(&prop, device_id));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            char msg[256];
            snprintf(msg, sizeof msg, "bpg_init: device %d is %s; this library is built for gfx950 (MI355X) only", device_id,
                     prop.gcnArchName);
            set_last_error_msg(msg);
            return BPG_EDEVICE;
        }
    }
    BPG_CHECK_HIP(hipSetDevice(device_id));
    // Host threads nap in stream / event waits instead of spinning (many prover threads per GPU share the host cores
    // with circuit capture and the transcript's Keccak chain).  BPG_SYNC = poll (default) | block | spin: see wait_event.
    const char *mode = getenv("BPG_SYNC");
    bpg_ctx *ctx = new bpg_ctx();
    ctx->device = device_id;
    ctx->id = g_next_ctx_id.fetch_add(1);
    ctx->sync_block = !(mode && !strcmp(mode, "spin"));
    ctx->sync_poll = true;
    if (mode && !strcmp(mode, "block")) {
        // refused once the device is in use with other flags: the library's own waits then keep polling
        unsigned flags = 0;
        (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
        (void)hipGetLastError();
        if (hipGetDeviceFlags(&flags) == hipSuccess && (flags & hipDeviceScheduleMask) == hipDeviceScheduleBlockingSync)
            ctx->sync_poll = false;
    }
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0)
            ctx->resident_lanes = cus * 4 * 4 * 64;
    }
    if (const char *s = getenv("BPG_TIME_KERNELS")) ctx->time_kernels = atoi(s) != 0;
    if (const char *s = getenv("BPG_MSM_C")) ctx->tune.c = atoi(s);
    if (const char *s = getenv("BPG_MSM_TILE")) ctx->tune.tile = atoi(s);
    if (const char *s = getenv("BPG_MSM_LS0")) ctx->tune.ls0 = atoi(s);
    if (const char *s = getenv("BPG_MSM_LS1")) ctx->tune.ls1 = atoi(s);
    if (const char *s = getenv("BPG_MSM_SORT_CHUNK")) ctx->tune.sort_chunk = atoi(s);
    if (const char *s = getenv("BPG_IPP_FOLD")) ctx->tune.ipp_fold = atoi(s);
    if (int rc = msm_noalloc_init(ctx)) {                  // address of this device's static workspace (bpg_msm_noalloc)
        delete ctx;
        return rc;
    }
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        g_live.push_back(ctx);
    }
    *out = ctx;
    return BPG_OK;
}

void bpg_shutdown(bpg_ctx *ctx) {
    if (!ctx) return;
    HostPhase::report();
    if (HostPhase::enabled())
        fprintf(stderr, "[bpg host phases] circuit-structure cache: %llu hits (%llu by following the cached circuit during capture), %llu misses\n",
                (unsigned long long)ctx->flat_hits.load(), (unsigned long long)ctx->flat_followed.load(), (unsigned long long)ctx->flat_misses.load());
    (void)hipSetDevice(ctx->device);
    // no call may be in flight on any thread; lanes of threads that have exited are freed here too
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        g_live.erase(std::remove(g_live.begin(), g_live.end(), ctx), g_live.end());
    }
    forget_shapes_of(ctx->id);
    for (auto &b : ctx->spec_pool)
        if (b.first) (void)hipHostFree(b.first);
    ctx->spec_pool.clear();
    for (auto *lane : ctx->lanes) {
        for (auto *j : lane->jobs) msm_job_destroy(j);
        (void)hipStreamSynchronize(lane->stream);
        if (lane->ws) (void)hipFree(lane->ws);
        if (lane->d_dyn) (void)hipFree(lane->d_dyn);
        if (lane->d_io) (void)hipFree(lane->d_io);
        if (lane->h_raw) (void)hipHostFree(lane->h_raw);
        if (lane->h_wit) (void)hipHostFree(lane->h_wit);
        if (lane->h_io) (void)hipHostFree(lane->h_io);
        if (lane->h_ped) (void)hipHostFree(lane->h_ped);
        if (lane->h_dec) (void)hipHostFree(lane->h_dec);
        if (lane->d_vacc) (void)hipFree(lane->d_vacc);
        if (lane->d_vtail) (void)hipFree(lane->d_vtail);
        if (lane->ev_sync) (void)hipEventDestroy(lane->ev_sync);
        for (auto &e : lane->ev_k)
            if (e) (void)hipEventDestroy(e);
        (void)hipStreamDestroy(lane->stream);
        delete lane;
    }
    delete ctx;
}

int bpg_device_alloc(bpg_ctx *ctx, size_t bytes, void **d_ptr) {
    if (!ctx || !d_ptr) return BPG_EINVAL;
    BPG_CHECK_HIP(hipSetDevice(ctx->device));
    BPG_CHECK_HIP(hipMalloc(d_ptr, bytes ? bytes : 16));
    return BPG_OK;
}
int bpg_device_free(bpg_ctx *ctx, void *d_ptr) {
    if (!ctx) return BPG_EINVAL;
    BPG_CHECK_HIP(hipFree(d_ptr));
    return BPG_OK;
}
int bpg_copy_to_device(bpg_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
    if (!ctx) return BPG_EINVAL;
    BPG_CHECK_HIP(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return BPG_OK;
}
int bpg_copy_to_host(bpg_ctx *ctx, void *h_dst, const void *d_src, size_t bytes) {
    if (!ctx) return BPG_EINVAL;
    BPG_CHECK_HIP(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return BPG_OK;
}
int bpg_synchronize(bpg_ctx *ctx) {
    if (!ctx) return BPG_EINVAL;
    BPG_CHECK_HIP(hipSetDevice(ctx->device));
    BPG_CHECK_HIP(hipDeviceSynchronize());
    return BPG_OK;
}

static int points_alloc(bpg_ctx *ctx, size_t n, bpg_points **out) {
    bpg_points *p = new bpg_points();
    p->ctx = ctx;
    p->n = n;
    p->d_table = nullptr;
    hipError_t e = hipMalloc((void **)&p->d_table, (n ? n : 1) * sizeof(aniels_rec));
    if (e != hipSuccess) {
        set_last_error("hipMalloc(point table)", e);
        delete p;
        return BPG_ENOMEM;
    }
    *out = p;
    return BPG_OK;
}

int bpg_points_from_compressed(bpg_ctx *ctx, const uint8_t *points32, size_t n, bpg_points **out) {
    if (!ctx || !out || (n && !points32)) return BPG_EINVAL;
    BPG_LANE(ctx);
    bpg_points *p = nullptr;
    int rc = points_alloc(ctx, n, &p);
    if (rc) return rc;
    if (n) {
        uint32_t bad = 0;
        if ((rc = table_write_compressed(ctx, p->d_table, points32, n, &bad))) {
            bpg_points_free(p);
            return rc;
        }
        if (bad) {
            bpg_points_free(p);
            set_last_error_msg("bpg_points_from_compressed: invalid ristretto255 encoding");
            return BPG_EFORMAT;
        }
    }
    *out = p;
    return BPG_OK;
}

int bpg_points_from_uniform(bpg_ctx *ctx, const uint8_t *uniform64, size_t n, bpg_points **out) {
    if (!ctx || !out || (n && !uniform64)) return BPG_EINVAL;
    BPG_LANE(ctx);
    bpg_points *p = nullptr;
    int rc = points_alloc(ctx, n, &p);
    if (rc) return rc;
    if (n && (rc = table_write_uniform(ctx, p->d_table, uniform64, n))) {
        bpg_points_free(p);
        return rc;
    }
    *out = p;
    return BPG_OK;
}

int bpg_points_compress(bpg_ctx *ctx, const bpg_points *pts, size_t start, size_t n, uint8_t *out32) {
    if (!ctx || !pts || (n && !out32) || start + n > pts->n) return BPG_EINVAL;
    if (!n) return BPG_OK;
    BPG_LANE(ctx);
    int rc;
    if ((rc = grow(&lane->d_io, &lane->io_cap, n * 32))) return rc;
    uint8_t *d_out = (uint8_t *)lane->d_io;
    hipLaunchKernelGGL(k_compress, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, lane->stream, pts->d_table + start, d_out, n);
    BPG_CHECK_HIP(hipGetLastError());
    BPG_CHECK_HIP(hipMemcpyAsync(out32, d_out, n * 32, hipMemcpyDeviceToHost, lane->stream));
    BPG_CHECK_HIP(lane_sync(lane));
    return BPG_OK;
}

size_t bpg_points_len(const bpg_points *pts) { return pts ? pts->n : 0; }

void bpg_points_free(bpg_points *pts) {
    if (!pts) return;
    (void)hipFree(pts->d_table);
    delete pts;
}

int bpg_msm_submit_segments(bpg_ctx *ctx, const bpg_points *pts, const size_t *seg_start, const size_t *seg_len,
                            size_t n_segments, const void *d_scalars, size_t n_targets, bpg_msm_job **job) {
    if (!seg_start || !seg_len || n_segments < 1 || n_segments > 4 || n_targets < 1 || n_targets > 65535) return BPG_EINVAL;
    if (!pts) return BPG_EINVAL;
    size_t n = 0;
    msm_seg segs[4];
    for (size_t k = 0; k < n_segments; k++) {
        if (seg_start[k] > pts->n || seg_len[k] > pts->n - seg_start[k]) return BPG_EINVAL;
        segs[k] = msm_seg{pts->d_table + seg_start[k], seg_len[k]};
        n += seg_len[k];
    }
    return msm_submit(ctx, segs, (int)n_segments, d_scalars, n, (int)n_targets, false, job);
}

int bpg_msm_submit(bpg_ctx *ctx, const bpg_points *pts, const void *d_scalars, size_t n, size_t n_targets, bpg_msm_job **job) {
    if (!pts || n > pts->n || n_targets < 1 || n_targets > 65535) return BPG_EINVAL;
    msm_seg seg{pts->d_table, n};
    return msm_submit(ctx, &seg, 1, d_scalars, n, (int)n_targets, false, job);
}

int bpg_msm_wait(bpg_msm_job *job, uint8_t *out32) { return msm_wait(job, out32, false); }
int bpg_msm_wait_partial(bpg_msm_job *job, uint8_t *out128) { return msm_wait(job, out128, true); }
int bpg_msm_combine_partials(const uint8_t *partials128, size_t k, uint8_t out32[32]) {
    if ((k && !partials128) || !out32) return BPG_EINVAL;
    return combine_partials(partials128, k, out32);
}

int bpg_msm(bpg_ctx *ctx, const bpg_points *pts, const uint8_t *scalars32, size_t n, uint8_t out32[32]) {
    if (!ctx || !pts || !out32 || (n && !scalars32) || n > pts->n) return BPG_EINVAL;
    msm_seg seg{pts->d_table, n};
    bpg_msm_job *job = nullptr;
    int rc = msm_submit(ctx, &seg, 1, scalars32, n, 1, true, &job);
    if (rc) {
        if (job) job->busy = false;
        return rc;
    }
    return msm_wait(job, out32, false);
}

int bpg_msm_noalloc(bpg_ctx *ctx, const bpg_points *pts, const void *d_scalars, size_t n, void *d_out32, void *stream) {
    if (!ctx || !pts || n > pts->n) return BPG_EINVAL;
    msm_seg seg{pts->d_table, n};
    return msm_noalloc(ctx, &seg, d_scalars, n, d_out32, (hipStream_t)stream);
}

int bpg_msm_plan(bpg_ctx *ctx, size_t n, size_t n_targets, int *window_bits, int *windows, int *buckets, int *threads_per_bucket) {
    if (!ctx) return BPG_EINVAL;
    msm_plan pl = make_plan(n, (int)n_targets, ctx->tune);
    if (window_bits) *window_bits = pl.c;
    if (windows) *windows = pl.W;
    if (buckets) *buckets = pl.NB;
    if (threads_per_bucket) *threads_per_bucket = pl.P;
    return BPG_OK;
}

void bpg_debug_fail_alloc(int countdown) { bpg::debug_fail_alloc(countdown); }

int bpg_msm_last_kernel_us(bpg_ctx *ctx, float *accumulate_us, float *total_device_us) {
    if (!ctx) return BPG_EINVAL;
    BPG_LANE(ctx);   // the calling thread's most recent bpg_msm_wait
    if (accumulate_us) *accumulate_us = lane->last_acc_us;
    if (total_device_us) *total_device_us = lane->last_total_us;
    return BPG_OK;
}

}  // extern "C"
