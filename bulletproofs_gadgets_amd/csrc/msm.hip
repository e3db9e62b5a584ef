// msm.hip -- Pippenger multiscalar multiplication over a resident affine-Niels
// table, hand-written for MI355X (gfx950, wave64, 160 KB LDS / CU).
//
// Replaces curve25519-dalek's Straus / Pippenger (RistrettoPoint::multiscalar_mul,
// vartime_multiscalar_mul, optional_multiscalar_mul) under Prover::prove and
// Verifier::verify (ref call sites: tests/edwards_point.rs:47,79; see include/bpg.h).
// The group result is canonical once compressed, so the schedule is free to differ
// from dalek's; parity is checked on the 32-byte output.
//
// Pipeline for n scalars x T targets, window c bits, W = 255/c+1 windows,
// NB = 2^(c-1) signed buckets per window:
//   k_digits      scalar -> W signed digits (int16), [target][window][i] layout   (HBM stream)
//   k_bin_hist / k_bin_scan / k_bin_scatter / k_chunk_count / k_chunk_offsets / k_chunk_place
//                 two-level bucket sort of the entries (point index | sign): bins, then buckets
//                 inside a bin with LDS cursors, a bin in chunks of bounded size so that skewed
//                 digits cannot pile a window onto one block; yields bucket counts and entry offsets
//   k_scan_a/b    exclusive scans: thread offsets of every pass
//   k_segmap      thread -> bucket maps of the accumulate pass and the first merge pass
//   k_accumulate  ceil(cnt / S0) threads per bucket: mixed additions (7M each)     (VALU int32: the hot kernel)
//   k_reduce      merge passes: S1 partials per thread until one sum per bucket is left
//   k_tail_a/b    an LDS binary tree that yields, per
//                 window, the bucket total and the c-1 "bit j of bucket index"
//                 sums in depth c-1 (sum_b (b+1) B_b = total + sum_j 2^j S_j)
//   host          Horner over the W*c bit positions (one 255-doubling chain, which a
//                 single GPU lane would take ~15x longer to walk) + ristretto compress.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <hip/hip_ext.h>

#include "bpg_internal.h"
#include "hot25519.h"

namespace bpg {

constexpr int MAX_LEVELS = 8;     // accumulate pass + up to 7 merge passes

// ------------------------------------------------------------------ device helpers

__device__ __forceinline__ void load_aniels(ge_aniels &q, const aniels_rec *rec) {
    const uint4 *p = reinterpret_cast<const uint4 *>(rec);
    uint4 a0 = p[0], a1 = p[1], b0 = p[2], b1 = p[3], c0 = p[4], c1 = p[5];
    q.hYpX.v[0] = a0.x; q.hYpX.v[1] = a0.y; q.hYpX.v[2] = a0.z; q.hYpX.v[3] = a0.w;
    q.hYpX.v[4] = a1.x; q.hYpX.v[5] = a1.y; q.hYpX.v[6] = a1.z; q.hYpX.v[7] = a1.w;
    q.hYmX.v[0] = b0.x; q.hYmX.v[1] = b0.y; q.hYmX.v[2] = b0.z; q.hYmX.v[3] = b0.w;
    q.hYmX.v[4] = b1.x; q.hYmX.v[5] = b1.y; q.hYmX.v[6] = b1.z; q.hYmX.v[7] = b1.w;
    q.Tdxy.v[0] = c0.x; q.Tdxy.v[1] = c0.y; q.Tdxy.v[2] = c0.z; q.Tdxy.v[3] = c0.w;
    q.Tdxy.v[4] = c1.x; q.Tdxy.v[5] = c1.y; q.Tdxy.v[6] = c1.z; q.Tdxy.v[7] = c1.w;
}

__device__ __forceinline__ void load_ge(ge &r, const ge_rec *rec) {
    const uint4 *p = reinterpret_cast<const uint4 *>(rec);
    fe *f[4] = {&r.X, &r.Y, &r.Z, &r.T};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint4 a = p[2 * k], b = p[2 * k + 1];
        f[k]->v[0] = a.x; f[k]->v[1] = a.y; f[k]->v[2] = a.z; f[k]->v[3] = a.w;
        f[k]->v[4] = b.x; f[k]->v[5] = b.y; f[k]->v[6] = b.z; f[k]->v[7] = b.w;
    }
}

__device__ __forceinline__ void store_ge(ge_rec *rec, const ge &r) {
    uint4 *p = reinterpret_cast<uint4 *>(rec);
    const fe *f[4] = {&r.X, &r.Y, &r.Z, &r.T};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        p[2 * k] = make_uint4(f[k]->v[0], f[k]->v[1], f[k]->v[2], f[k]->v[3]);
        p[2 * k + 1] = make_uint4(f[k]->v[4], f[k]->v[5], f[k]->v[6], f[k]->v[7]);
    }
}

// SoA LDS image of extended points: word k of slot s at lds[k * nslots + s]
__device__ __forceinline__ void lds_put(uint32_t *lds, int nslots, int s, const ge &p) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        lds[i * nslots + s] = p.X.v[i];
        lds[(8 + i) * nslots + s] = p.Y.v[i];
        lds[(16 + i) * nslots + s] = p.Z.v[i];
        lds[(24 + i) * nslots + s] = p.T.v[i];
    }
}
__device__ __forceinline__ void lds_get(ge &p, const uint32_t *lds, int nslots, int s) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        p.X.v[i] = lds[i * nslots + s];
        p.Y.v[i] = lds[(8 + i) * nslots + s];
        p.Z.v[i] = lds[(16 + i) * nslots + s];
        p.T.v[i] = lds[(24 + i) * nslots + s];
    }
}

// Binary tree over bucket-index bits, held in LDS.  State: `nodes` nodes, each
// with `per` points [total, S_0 .. S_{per-2}].  Merging siblings L (bit = 0), R (bit = 1):
//   total' = total_L + total_R,  S_j' = S_j,L + S_j,R,  S_new = total_R.
// After `levels` merges one node remains with per + levels points in slots 0...
__device__ void lds_bit_tree(uint32_t *lds, int nslots, int nodes, int per, int levels) {
    const int tid = threadIdx.x;
    for (int l = 0; l < levels; l++) {
        const int out_per = per + 1, out_nodes = nodes >> 1, total = out_nodes * out_per;
        ge o;
        bool active = tid < total;
        if (active) {
            int node = tid / out_per, j = tid - node * out_per;
            int L = 2 * node, R = L + 1;
            if (j < per) {
                ge a, b;
                lds_get(a, lds, nslots, L * per + j);
                lds_get(b, lds, nslots, R * per + j);
                ge_add(o, a, b);               // (hot25519's gez_add here: 140 B of spills, 110 -> 124 us)
            } else {
                lds_get(o, lds, nslots, R * per);
            }
        }
        __syncthreads();
        if (active) lds_put(lds, nslots, tid, o);
        __syncthreads();
        per = out_per;
        nodes = out_nodes;
    }
}

// ------------------------------------------------------------------ kernels

// windows needed for a scalar below 2^251 + 2^125 (any scalar after the mod-l reduction and
// the negation below): ceil(252 / c).  With c | 252 (c = 14, 12, 9, 7, 6, 4) every window,
// the top one included, sees uniform digits.
__host__ __device__ constexpr int windows_for(int c) { return (252 + c - 1) / c; }

// Where scalar i of a target's row lives: segment k covers [end[k-1], end[k]) of the MSM's
// index space and starts off[k] scalars into the row; rows are `stride` scalars apart.
// (Dense rows: off[k] = end[k-1], stride = n.  A rank of a point-range-sharded MSM reads
// its slices of the full-length rows in place.)
struct sc_map {
    uint32_t end[4];
    uint32_t off[4];
    size_t stride;
};

// scalar -> signed c-bit digits.  digits[(t*W + w)*n + i]
template <int C>
__global__ void __launch_bounds__(256) k_digits(const uint32_t *__restrict__ scalars, int16_t *__restrict__ digits,
                                                size_t n, size_t total, sc_map sm, uint32_t *__restrict__ zero, uint32_t nzero) {
    constexpr int W = windows_for(C);
    size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    // clears the sort's bin counters on the way (a separate memset is up to three fill kernels)
    for (size_t z = gid; z < nzero; z += (size_t)gridDim.x * 256) zero[z] = 0u;
    if (gid >= total) return;
    size_t t = gid / n, i = gid - t * n;
    uint32_t start = 0, off = sm.off[0];
#pragma unroll
    for (int k = 1; k < 4; k++) {
        const bool past = i >= sm.end[k - 1];
        start = past ? sm.end[k - 1] : start;
        off = past ? sm.off[k] : off;
    }
    const size_t src = t * sm.stride + off + (i - start);
    const uint4 *sp = reinterpret_cast<const uint4 *>(scalars + src * 8);
    uint4 lo4 = sp[0], hi4 = sp[1];
    uint32_t v[9] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w, 0u};
    const uint32_t HL[8] = {0x2e7ae9f6u, 0x2c09318du, 0x517bce6bu, 0x0a6f7cefu, 0u, 0u, 0u, 0x08000000u};
    const uint32_t LL[8] = {0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0u, 0u, 0u, 0x10000000u};
    // any 256-bit input is taken mod l (2^256 < 16 l): subtract 8l, 4l, 2l, l where they fit.
    // The points have order l up to 4-torsion, which the Ristretto encoding ignores.
#pragma unroll
    for (int sh = 3; sh >= 0; sh--) {
        uint32_t m[8], d[8], b = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) m[k] = (LL[k] << sh) | ((sh && k) ? (LL[k - 1] >> (32 - sh)) : 0u);
#pragma unroll
        for (int k = 0; k < 8; k++) d[k] = subb32(v[k], m[k], b, b);
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = b ? v[k] : d[k];
    }
    // s > (l-1)/2  ->  use l - s and flip every digit's sign (s*P = (l-s)*(-P)).
    // Keeps s at or below 2^251 so the signed recoding never carries out of the top
    // window: for uniform scalars that carry would put ~n/2 entries in one bucket.
    uint32_t bw = 0, nb = 0, ng[8], dummy;
#pragma unroll
    for (int k = 0; k < 8; k++) dummy = subb32(HL[k], v[k], bw, bw);     // bw = (s > half)
    (void)dummy;
#pragma unroll
    for (int k = 0; k < 8; k++) ng[k] = subb32(LL[k], v[k], nb, nb);
    const bool flip = bw != 0;
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = flip ? ng[k] : v[k];
    int16_t *out = digits + (t * W) * n + i;
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < W; w++) {
        constexpr uint32_t mask = (1u << C) - 1u, half = 1u << (C - 1);
        const int bit = w * C, word = bit >> 5, off = bit & 31;
        uint32_t raw = 0;
        if (word < 8) {
            uint64_t two = ((uint64_t)v[word + 1] << 32) | v[word];
            raw = (uint32_t)(two >> off) & mask;
        }
        raw += carry;
        // raw == half stays (no carry), which is what keeps the top window from carrying out when c divides 252.
        // At c = 16 a stored digit must also fit int16: the tie then carries when the digit is stored as it is
        // (d = -2^15 instead of +2^15; stored negated, +2^15 becomes -2^15).  The top window of c = 16 holds 12 bits:
        // no tie there.
        carry = (raw > half || (C == 16 && !flip && raw == half)) ? 1u : 0u;
        int d = (int)raw - (int)(carry << C);
        out[(size_t)w * n] = (int16_t)(flip ? -d : d);
    }
}

// ---- bucket sort of the non-zero digits, in two levels -------------------------------------
// A bucket set (one per (target, window); fixed-base mode: one per target, shared by all its
// windows) has NB = NBIN * LOWN buckets.  Level 1 splits the entries of a set into NBIN bins by
// the high bits of the bucket number: a block owns one (tile, window, target), keeps its bin
// cursors in LDS and claims a run in each bin's region with one global atomic, so its stores
// land in NBIN runs of ~tile/NBIN entries.  Level 2 gives each bin to one block, which counts
// the LOWN buckets of the bin in LDS, writes counts and entry offsets of its buckets, and
// places the entries: all its stores fall into the bin's own region (tens of KB), which the
// L2 holds until every line is complete.  (A one-level scatter to 8192 buckets wrote 16-byte
// runs all over a 75 MB array: measured 7.7x write amplification and 4.7 M returning atomics.)

constexpr int SORT_U = 8;

// Level-1 output item: (entry, bucket number within its bin).  A variable-base MSM of at most 2^23 points packs it
// into one word -- point index (23 bits) | bucket in bin (8 bits) | sign -- which takes a third off the sort's
// traffic (the item is written once and read twice); otherwise it is a pair of words.
template <bool PACKED> struct sort_item;
template <> struct sort_item<false> {
    using T = uint2;
    static __device__ __forceinline__ T make(uint32_t entry, uint32_t low) { return make_uint2(entry, low); }
    static __device__ __forceinline__ uint32_t low(const T &t) { return t.y; }
    static __device__ __forceinline__ uint32_t entry(const T &t) { return t.x; }
};
template <> struct sort_item<true> {
    using T = uint32_t;
    static __device__ __forceinline__ T make(uint32_t entry, uint32_t low) { return entry | (low << 23); }
    static __device__ __forceinline__ uint32_t low(const T &t) { return (t >> 23) & 0xffu; }
    static __device__ __forceinline__ uint32_t entry(const T &t) { return t & 0x807fffffu; }
};
constexpr size_t SORT_PACK_MAX_POINTS = (size_t)1 << 23;

// LDS cursor increment, aggregated when every active lane of the wave asks for the same counter
// (skewed digits: bit witnesses send a whole tile to one bin / one bucket; 64 lanes hammering
// one LDS address serialise).  Returns the lane's slot (old counter value + its rank among the
// wave's active lanes).  Wave-uniform control flow; inactive lanes get 0.
__device__ __forceinline__ uint32_t lds_claim(uint32_t *ctr, uint32_t key, bool active) {
    const uint64_t act = __ballot(active);
    if (!act) return 0u;
    const int first = __ffsll((unsigned long long)act) - 1;
    const uint32_t k0 = __shfl(key, first, 64);
    const uint64_t same = __ballot(active && key == k0);
    if (same == act) {
        const int lane = threadIdx.x & 63;
        uint32_t old = 0;
        if (lane == first) old = atomicAdd(&ctr[k0], (uint32_t)__popcll(same));
        old = __shfl(old, first, 64);
        return old + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
    }
    return active ? atomicAdd(&ctr[key], 1u) : 0u;
}

// bin histogram: bin_counts[bset * NBIN + bin] += entries of this (tile, window, target)
__global__ void __launch_bounds__(512) k_bin_hist(const int16_t *__restrict__ digits, uint32_t *__restrict__ bin_counts, size_t n,
                                                  int W, int NBIN, int lowshift, int tile, int shared) {
    extern __shared__ uint32_t lds[];
    const int w = blockIdx.y, t = blockIdx.z;
    const size_t base = (size_t)t * W + w, bset = shared ? (size_t)t : base;
    const int16_t *dg = digits + base * n;
    for (int b = threadIdx.x; b < NBIN; b += blockDim.x) lds[b] = 0;
    __syncthreads();
    size_t start = (size_t)blockIdx.x * tile, end = start + tile < n ? start + tile : n;
    // SORT_U independent loads per step: these loops are bound by load latency, not by bytes
    for (size_t i0 = start + threadIdx.x; i0 < end; i0 += (size_t)blockDim.x * SORT_U) {
        int d[SORT_U];
#pragma unroll
        for (int u = 0; u < SORT_U; u++) {
            const size_t i = i0 + (size_t)u * blockDim.x;
            d[u] = i < end ? dg[i] : 0;
        }
#pragma unroll
        for (int u = 0; u < SORT_U; u++)
            (void)lds_claim(lds, (uint32_t)((d[u] < 0 ? -d[u] : d[u]) - 1) >> lowshift, d[u] != 0);
    }
    __syncthreads();
    uint32_t *gc = bin_counts + bset * NBIN;
    for (int b = threadIdx.x; b < NBIN; b += blockDim.x) {
        uint32_t c = lds[b];
        if (c) atomicAdd(&gc[b], c);
    }
}

// exclusive scans of the bin counts and of the bins' chunk counts (one block; at most a few thousand
// bins).  Level 2 works on chunks of at most CH items of one bin, so a bin that a skewed digit
// distribution filled (bit witnesses: half a window's entries in one bin) is spread over many blocks.
__global__ void __launch_bounds__(1024) k_bin_scan(const uint32_t *__restrict__ bin_counts, uint32_t *__restrict__ bin_off,
                                                   uint32_t *__restrict__ chunk_off, uint32_t nbt, uint32_t CH) {
    __shared__ uint32_t wsum[16], wsumc[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // thread i owns elements [i * per, (i + 1) * per)
    const uint32_t per = (nbt + 1023u) / 1024u, first = threadIdx.x * per;
    uint32_t mine = 0, minec = 0;
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t i = first + k;
        if (i < nbt) {
            const uint32_t v = bin_counts[i];
            mine += v;
            minec += v > CH ? (v + CH - 1) / CH : 1u;      // an empty bin keeps one chunk: it writes the zero counts
        }
    }
    uint32_t x = mine, xc = minec;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t y = __shfl_up(x, o, 64), yc = __shfl_up(xc, o, 64);
        if (lane >= o) {
            x += y;
            xc += yc;
        }
    }
    if (lane == 63) {
        wsum[wave] = x;
        wsumc[wave] = xc;
    }
    __syncthreads();
    uint32_t run = x - mine, runc = xc - minec;
    for (int k = 0; k < wave; k++) {
        run += wsum[k];
        runc += wsumc[k];
    }
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t i = first + k;
        if (i < nbt) {
            const uint32_t v = bin_counts[i];
            bin_off[i] = run;
            chunk_off[i] = runc;
            run += v;
            runc += v > CH ? (v + CH - 1) / CH : 1u;
        }
    }
    if (threadIdx.x == 1023) {                             // its running sums are the totals (elements past nbt add nothing)
        bin_off[nbt] = run;
        chunk_off[nbt] = runc;
    }
}

// level 1: entries -> bins.  tmp[pos] = (entry, bucket number within the bin); entry = scalar
// index | sign << 31 (fixed-base mode: also the window << FB_WSHIFT, i.e. which block of the shifted
// table the point comes from: 25 index bits, 6 window bits)
constexpr int FB_WSHIFT = 25;
constexpr uint32_t FB_IDX_MASK = (1u << FB_WSHIFT) - 1u, FB_WMASK = 63u;
template <bool PACKED>
__global__ void __launch_bounds__(512) k_bin_scatter(const int16_t *__restrict__ digits, const uint32_t *__restrict__ bin_off,
                                                     uint32_t *__restrict__ bin_fill, typename sort_item<PACKED>::T *__restrict__ tmp,
                                                     size_t n, int W, int NBIN, int lowshift, int tile, int shared) {
    extern __shared__ uint32_t lds[];
    const int w = blockIdx.y, t = blockIdx.z;
    const size_t base = (size_t)t * W + w, bset = shared ? (size_t)t : base;
    const uint32_t wtag = shared ? (uint32_t)w << FB_WSHIFT : 0u;
    const uint32_t lowmask = (1u << lowshift) - 1u;
    const int16_t *dg = digits + base * n;
    for (int b = threadIdx.x; b < NBIN; b += blockDim.x) lds[b] = 0;
    __syncthreads();
    size_t start = (size_t)blockIdx.x * tile, end = start + tile < n ? start + tile : n;
    for (size_t i0 = start + threadIdx.x; i0 < end; i0 += (size_t)blockDim.x * SORT_U) {
        int d[SORT_U];
#pragma unroll
        for (int u = 0; u < SORT_U; u++) {
            const size_t i = i0 + (size_t)u * blockDim.x;
            d[u] = i < end ? dg[i] : 0;
        }
#pragma unroll
        for (int u = 0; u < SORT_U; u++)
            (void)lds_claim(lds, (uint32_t)((d[u] < 0 ? -d[u] : d[u]) - 1) >> lowshift, d[u] != 0);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < NBIN; b += blockDim.x) {
        uint32_t c = lds[b];
        if (c) {
            size_t gb = bset * NBIN + b;
            lds[b] = bin_off[gb] + atomicAdd(&bin_fill[gb], c);
        }
    }
    __syncthreads();
    for (size_t i0 = start + threadIdx.x; i0 < end; i0 += (size_t)blockDim.x * SORT_U) {
        int d[SORT_U];
#pragma unroll
        for (int u = 0; u < SORT_U; u++) {
            const size_t i = i0 + (size_t)u * blockDim.x;
            d[u] = i < end ? dg[i] : 0;
        }
#pragma unroll
        for (int u = 0; u < SORT_U; u++) {
            const size_t i = i0 + (size_t)u * blockDim.x;
            const uint32_t bk = (uint32_t)(d[u] < 0 ? -d[u] : d[u]) - 1u;
            const uint32_t pos = lds_claim(lds, bk >> lowshift, d[u] != 0);
            if (d[u] != 0) tmp[pos] = sort_item<PACKED>::make((uint32_t)i | wtag | (d[u] < 0 ? 0x80000000u : 0u), bk & lowmask);
        }
    }
}

// level 2, chunk c -> its bin and its items [lo, hi) of tmp.  false: c is past the last chunk.
__device__ __forceinline__ bool chunk_range(const uint32_t *__restrict__ bin_off, const uint32_t *__restrict__ chunk_off,
                                            uint32_t nbt, uint32_t CH, uint32_t c, uint32_t &bin, uint32_t &lo, uint32_t &hi) {
    const uint32_t nch = chunk_off[nbt];
    if (c >= nch) return false;
    uint32_t a = c, b = nbt;                               // every bin a single chunk (uniform digits): chunk c is bin c
    if (nch != nbt) {
        a = 0;                                             // chunk_off[a] <= c < chunk_off[b]
        while (b - a > 1) {
            const uint32_t m = (a + b) >> 1;
            if (chunk_off[m] <= c) a = m;
            else b = m;
        }
    }
    bin = a;
    lo = bin_off[a] + (c - chunk_off[a]) * CH;
    const uint32_t end = bin_off[a + 1];
    hi = end - lo > CH ? lo + CH : end;
    return true;
}

// level 2a: one block per chunk counts the chunk's items per bucket of its bin: chunk_hist[c][LOWN].
// A bin that is a single chunk (every bin, when the digits are uniform) is finished here: counts[g],
// ent_off[g], entries sorted by bucket; 2b and 2c then skip it.
template <bool PACKED>
__global__ void __launch_bounds__(256) k_chunk_count(const uint32_t *__restrict__ bin_off, const uint32_t *__restrict__ chunk_off,
                                                     const typename sort_item<PACKED>::T *__restrict__ tmp, uint32_t *__restrict__ chunk_hist,
                                                     uint32_t *__restrict__ counts, uint32_t *__restrict__ ent_off,
                                                     uint32_t *__restrict__ entries, uint32_t nbt, uint32_t CH, int LOWN) {
    __shared__ uint32_t cnt[256], cur[256];
    __shared__ uint32_t wsum[4];
    uint32_t bin, lo, hi;
    if (!chunk_range(bin_off, chunk_off, nbt, CH, blockIdx.x, bin, lo, hi)) return;
    const int tid = threadIdx.x;
    cnt[tid] = 0;
    __syncthreads();
    for (uint32_t i0 = lo + tid; i0 < hi; i0 += 256u * SORT_U) {
        uint32_t b[SORT_U];
#pragma unroll
        for (int u = 0; u < SORT_U; u++) {
            const uint32_t i = i0 + 256u * u;
            b[u] = i < hi ? sort_item<PACKED>::low(tmp[i]) : 0xffffffffu;
        }
#pragma unroll
        for (int u = 0; u < SORT_U; u++) (void)lds_claim(cnt, b[u], b[u] != 0xffffffffu);
    }
    __syncthreads();
    if (chunk_off[bin + 1] - chunk_off[bin] != 1u) {       // block-uniform
        if (tid < LOWN) chunk_hist[(size_t)blockIdx.x * LOWN + tid] = cnt[tid];
        return;
    }
    // exclusive scan of cnt[0..256) (LOWN <= 256; the rest is zero)
    const uint32_t mine = cnt[tid];
    uint32_t x = mine;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint32_t wbase = 0;
    for (int k = 0; k < wave; k++) wbase += wsum[k];
    const uint32_t off = lo + wbase + x - mine;
    cur[tid] = off;
    if (tid < LOWN) {
        const size_t g = (size_t)bin * LOWN + tid;
        counts[g] = mine;
        ent_off[g] = off;
    }
    __syncthreads();
    for (uint32_t i0 = lo + tid; i0 < hi; i0 += 256u * SORT_U) {
        uint32_t lw[SORT_U], en[SORT_U];
#pragma unroll
        for (int u = 0; u < SORT_U; u++) {
            const uint32_t i = i0 + 256u * u;
            lw[u] = 0xffffffffu;
            en[u] = 0u;
            if (i < hi) {
                const typename sort_item<PACKED>::T t = tmp[i];
                lw[u] = sort_item<PACKED>::low(t);
                en[u] = sort_item<PACKED>::entry(t);
            }
        }
#pragma unroll
        for (int u = 0; u < SORT_U; u++) {
            const uint32_t pos = lds_claim(cur, lw[u], lw[u] != 0xffffffffu);
            if (lw[u] != 0xffffffffu) entries[pos] = en[u];
        }
    }
}

// level 2b: one block per bin.  counts[g], ent_off[g] for the bin's LOWN buckets; chunk_hist becomes
// each chunk's first slot within the bucket (exclusive running sum over the bin's chunks).
__global__ void __launch_bounds__(256) k_chunk_offsets(const uint32_t *__restrict__ bin_off, const uint32_t *__restrict__ chunk_off,
                                                       uint32_t *__restrict__ chunk_hist, uint32_t *__restrict__ counts,
                                                       uint32_t *__restrict__ ent_off, uint32_t nbt, int LOWN) {
    __shared__ uint32_t wsum[4];
    if (chunk_off[nbt] == nbt) return;                     // every bin is a single chunk: 2a did it all
    const int tid = threadIdx.x;
    for (uint32_t bin = blockIdx.x; bin < nbt; bin += gridDim.x) {
        const uint32_t lo = bin_off[bin], c0 = chunk_off[bin], c1 = chunk_off[bin + 1];
        if (c1 - c0 == 1u) continue;
        uint32_t mine = 0;
        if (tid < LOWN) {
            for (uint32_t c = c0; c < c1; c += SORT_U) {
                uint32_t v[SORT_U];
#pragma unroll
                for (int u = 0; u < SORT_U; u++) v[u] = c + u < c1 ? chunk_hist[(size_t)(c + u) * LOWN + tid] : 0u;
#pragma unroll
                for (int u = 0; u < SORT_U; u++) {
                    if (c + u < c1) chunk_hist[(size_t)(c + u) * LOWN + tid] = mine;
                    mine += v[u];
                }
            }
        }
        // exclusive scan over the bin's buckets (LOWN <= 256; the rest is zero)
        uint32_t x = mine;
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint32_t wbase = 0;
        for (int k = 0; k < wave; k++) wbase += wsum[k];
        if (tid < LOWN) {
            const size_t g = (size_t)bin * LOWN + tid;
            counts[g] = mine;
            ent_off[g] = lo + wbase + x - mine;
        }
        __syncthreads();                                   // wsum is reused
    }
}

// level 2c: one block per chunk places the chunk's entries: entries sorted by bucket
template <bool PACKED>
__global__ void __launch_bounds__(256) k_chunk_place(const uint32_t *__restrict__ bin_off, const uint32_t *__restrict__ chunk_off,
                                                     const typename sort_item<PACKED>::T *__restrict__ tmp, const uint32_t *__restrict__ chunk_hist,
                                                     const uint32_t *__restrict__ ent_off, uint32_t *__restrict__ entries,
                                                     uint32_t nbt, uint32_t CH, int LOWN) {
    __shared__ uint32_t cur[256];
    const uint32_t nch = chunk_off[nbt];
    if (nch == nbt) return;                                // every bin is a single chunk: 2a did it all
    const int tid = threadIdx.x;
    for (uint32_t c = blockIdx.x; c < nch; c += gridDim.x) {
        uint32_t bin, lo, hi;
        (void)chunk_range(bin_off, chunk_off, nbt, CH, c, bin, lo, hi);
        if (chunk_off[bin + 1] - chunk_off[bin] == 1u) continue;
        cur[tid] = tid < LOWN ? ent_off[(size_t)bin * LOWN + tid] + chunk_hist[(size_t)c * LOWN + tid] : 0u;
        __syncthreads();
        for (uint32_t i0 = lo + tid; i0 < hi; i0 += 256u * SORT_U) {
            uint32_t lw[SORT_U], en[SORT_U];
#pragma unroll
            for (int u = 0; u < SORT_U; u++) {
                const uint32_t i = i0 + 256u * u;
                lw[u] = 0xffffffffu;
                en[u] = 0u;
                if (i < hi) {
                    const typename sort_item<PACKED>::T t = tmp[i];
                    lw[u] = sort_item<PACKED>::low(t);
                    en[u] = sort_item<PACKED>::entry(t);
                }
            }
#pragma unroll
            for (int u = 0; u < SORT_U; u++) {
                const uint32_t pos = lds_claim(cur, lw[u], lw[u] != 0xffffffffu);
                if (lw[u] != 0xffffffffu) entries[pos] = en[u];
            }
        }
        __syncthreads();                                   // cur is reused
    }
}

// Segment counts.  A bucket with `cnt` entries is summed by ceil(cnt / S0)
// threads (level 0), whose partial sums are merged S1 at a time by the
// reduction passes (levels 1..R) until one sum per bucket is left.  This keeps
// every thread's work bounded whatever the digit distribution is: the signed
// recoding sends ~n/2 carries to one bucket of the top window, and gadget
// witnesses (bits, small values) pile up in a few low buckets.
//   level -1: cnt            level 0: ceil(cnt/S0)           level r: ceil(level r-1 / S1)
__device__ __forceinline__ uint32_t seg_count(uint32_t cnt, int level, int ls0, int ls1) {
    if (level < 0) return cnt;
    uint32_t x = (cnt + (uint32_t)ls0 - 1) / (uint32_t)ls0;      // ls0: S0, the segment size (any integer >= 1)
    for (int i = 0; i < level; i++) x = (x + (1u << ls1) - 1) >> ls1;
    return x;
}
// number of threads a bucket contributes to pass `level`: passes >= 1 only touch
// buckets that still have more than one partial
__device__ __forceinline__ uint32_t pass_threads(uint32_t cnt, int level, int ls0, int ls1) {
    if (level <= 0) return seg_count(cnt, level, ls0, ls1);
    uint32_t prev = seg_count(cnt, level - 1, ls0, ls1);
    return prev > 1 ? ((prev + (1u << ls1) - 1) >> ls1) : 0u;
}

// exclusive scans of pass_threads(counts[i], level) for ALL levels -1 .. nlev-2 in two
// launches (every level depends on the counts only).  Level array idx (= level + 1) lives at
// lvl + idx * lvl_words: local[ne] | base[nblk + 1].
// phase A: 1024 elements per block -> offsets local to the block + block total.
__global__ void __launch_bounds__(256) k_scan_a(const uint32_t *__restrict__ counts, uint32_t *__restrict__ lvl, size_t lvl_words,
                                                size_t ne, int nlev, int ls0, int ls1) {
    __shared__ uint32_t wsum[MAX_LEVELS + 1][4];
    const size_t base = (size_t)blockIdx.x * 1024 + (size_t)threadIdx.x * 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t cnt[4];
#pragma unroll
    for (int k = 0; k < 4; k++) cnt[k] = (base + k < ne) ? counts[base + k] : 0u;
    uint32_t x[MAX_LEVELS + 1], mine[MAX_LEVELS + 1];
#pragma unroll
    for (int idx = 0; idx < MAX_LEVELS + 1; idx++) {
        if (idx >= nlev) break;
        uint32_t m = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) m += pass_threads(cnt[k], idx - 1, ls0, ls1);
        uint32_t v = m;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t y = __shfl_up(v, o, 64);
            if (lane >= o) v += y;
        }
        if (lane == 63) wsum[idx][wave] = v;
        x[idx] = v;
        mine[idx] = m;
    }
    __syncthreads();
#pragma unroll
    for (int idx = 0; idx < MAX_LEVELS + 1; idx++) {
        if (idx >= nlev) break;
        uint32_t *local = lvl + (size_t)idx * lvl_words, *totals = local + ne;
        uint32_t wbase = 0;
        for (int k = 0; k < wave; k++) wbase += wsum[idx][k];
        uint32_t excl = wbase + x[idx] - mine[idx];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (base + k < ne) local[base + k] = excl;
            excl += pass_threads(cnt[k], idx - 1, ls0, ls1);
        }
        if (threadIdx.x == 255) totals[blockIdx.x] = wbase + x[idx];
    }
}

// phase B: block totals -> block bases, one block per level; totals[nblk] = grand total.
__global__ void __launch_bounds__(1024) k_scan_b(uint32_t *__restrict__ lvl, size_t lvl_words, size_t ne, size_t nblk) {
    uint32_t *totals = lvl + (size_t)blockIdx.x * lvl_words + ne;
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t running;
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (size_t start = 0; start < nblk; start += 1024) {
        size_t i = start + threadIdx.x;
        uint32_t mine = i < nblk ? totals[i] : 0u;
        uint32_t x = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint32_t wbase = running;
        for (int k = 0; k < wave; k++) wbase += wsum[k];
        if (i < nblk) totals[i] = wbase + x - mine;
        __syncthreads();
        if (threadIdx.x == 1023) running = wbase + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[nblk] = running;
}

struct seg_map {
    uint32_t len[4];             // cumulative ends of the segments in scalar-index space
    const aniels_rec *base[4];   // record of scalar index i in segment k is base[k][i]  (base pre-shifted by -len[k-1])
};

__device__ __forceinline__ const aniels_rec *map_record(const seg_map &m, uint32_t i) {
    const aniels_rec *b = m.base[0];
    b = i >= m.len[0] ? m.base[1] : b;
    b = i >= m.len[1] ? m.base[2] : b;
    b = i >= m.len[2] ? m.base[3] : b;
    return b + i;
}

// One scanned level: offsets = local[g] + base[g >> 10]; base[nblk] = total.
struct level_ref {
    const uint32_t *local;
    const uint32_t *base;
};

// bucket g with off(g) <= t < off(g+1), by a two-level binary search
__device__ __forceinline__ uint32_t find_bucket(const level_ref &lv, uint32_t t, uint32_t nblk, uint32_t ne, uint32_t &rank) {
    uint32_t lo = 0, hi = nblk;              // first block j with base[j] > t
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (lv.base[mid] > t) hi = mid; else lo = mid + 1;
    }
    uint32_t j = lo - 1;
    uint32_t tt = t - lv.base[j];
    uint32_t s = j << 10, e = s + 1024 < ne ? s + 1024 : ne;
    lo = s; hi = e;                          // first g in block with local[g] > tt
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (lv.local[mid] > tt) hi = mid; else lo = mid + 1;
    }
    uint32_t g = lo - 1;
    rank = tt - lv.local[g];
    return g;
}

// Thread -> bucket maps of the accumulate pass (level 0) and of the first merge pass (level 1):
// map[off(g) + r] = g for every thread r of bucket g.  A thread of those passes then finds its
// bucket with ONE coalesced load instead of an 18-step binary search, i.e. a chain of 18
// dependent memory round trips that every wave of a freshly launched round walked at the same
// time while its SIMD sat idle.  One thread per bucket; buckets with more than 32 threads
// (skewed digits) are finished by the whole wave, 64 entries per step.
__global__ void __launch_bounds__(256) k_segmap(const uint32_t *__restrict__ counts, level_ref lv0, level_ref lv1,
                                                uint32_t *__restrict__ map0, uint32_t *__restrict__ map1, uint32_t ne, int ls0,
                                                int ls1) {
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const uint32_t cnt = g < ne ? counts[g] : 0u;
#pragma unroll
    for (int level = 0; level < 2; level++) {
        if (level == 1 && !map1) break;
        const level_ref &lv = level ? lv1 : lv0;
        uint32_t *map = level ? map1 : map0;
        const uint32_t ns = pass_threads(cnt, level, ls0, ls1);
        const uint32_t off = ns ? lv.local[g] + lv.base[g >> 10] : 0u;
        const uint32_t first = ns < 32u ? ns : 32u;
        for (uint32_t r = 0; r < first; r++) map[off + r] = g;
        uint64_t big = __ballot(ns > 32u);
        while (big) {
            const int src = __ffsll((unsigned long long)big) - 1;
            big &= big - 1;
            const uint32_t bg = __shfl(g, src, 64), boff = __shfl(off, src, 64), bns = __shfl(ns, src, 64);
            for (uint32_t r = 32u + lane; r < bns; r += 64u) map[boff + r] = bg;
        }
    }
}

// record of scalar index i.  The four segment bases are wave-uniform kernel arguments: they are
// pinned in SGPRs so that the choice is three v_cndmask pairs -- left alone, the compiler turns the
// select chain into an indexed load from the kernarg segment, i.e. one more dependent memory
// round trip in front of every point gather.
template <bool MULTI>
__device__ __forceinline__ const aniels_rec *record_of(const seg_map &m, uint32_t i) {
    uint64_t b0 = (uint64_t)m.base[0];
    if (MULTI) {
        uint64_t b1 = (uint64_t)m.base[1], b2 = (uint64_t)m.base[2], b3 = (uint64_t)m.base[3];
        asm volatile("" : "+s"(b0), "+s"(b1), "+s"(b2), "+s"(b3));
        b0 = i >= m.len[0] ? b1 : b0;
        b0 = i >= m.len[1] ? b2 : b0;
        b0 = i >= m.len[2] ? b3 : b0;
    }
    return (const aniels_rec *)b0 + i;
}

// The table entry as the digit's sign selects it: a record is ((y+x)/2 | (y-x)/2 | dxy | -dxy), 32 bytes each, so the
// entry of -P is the same record read in another order -- three addresses instead of 48 v_cndmask_b32.
__device__ __forceinline__ void load_aniels_signed(ge_aniels_signed &q, const aniels_rec *rec, uint32_t neg) {
    const uint4 *p = reinterpret_cast<const uint4 *>(rec);
    const uint4 *pa = p + 2 * neg, *pb = p + 2 - 2 * neg, *pc = p + 4 + 2 * neg;
    uint4 a0 = pa[0], a1 = pa[1], b0 = pb[0], b1 = pb[1], c0 = pc[0], c1 = pc[1];
    q.A.v[0] = a0.x; q.A.v[1] = a0.y; q.A.v[2] = a0.z; q.A.v[3] = a0.w;
    q.A.v[4] = a1.x; q.A.v[5] = a1.y; q.A.v[6] = a1.z; q.A.v[7] = a1.w;
    q.B.v[0] = b0.x; q.B.v[1] = b0.y; q.B.v[2] = b0.z; q.B.v[3] = b0.w;
    q.B.v[4] = b1.x; q.B.v[5] = b1.y; q.B.v[6] = b1.z; q.B.v[7] = b1.w;
    q.C.v[0] = c0.x; q.C.v[1] = c0.y; q.C.v[2] = c0.z; q.C.v[3] = c0.w;
    q.C.v[4] = c1.x; q.C.v[5] = c1.y; q.C.v[6] = c1.z; q.C.v[7] = c1.w;
}

// THE hot kernel.  Thread t owns segment `rank` (about S0 entries) of bucket g = map0[t] and sums it with mixed
// additions (7M each), the generated instruction stream of madd_gfx950.inc: VALU-issue bound, 504 v_mad_u64_u32
// and ~660 other VALU instructions per addition, four waves per SIMD (128 VGPRs: 32 accumulator, 24 table entry,
// 56 scratch of the stream).  The first entry of a segment costs one multiplication, not seven: identity + P = P.
// No LDS staging: every lane adds a different table entry into a different partial sum, so there is nothing for
// the lanes of a workgroup to share; the entries of a bucket are contiguous and the table is gathered by 128-byte
// records, one cache line each.
template <bool MULTI>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_accumulate(const uint32_t *__restrict__ entries, const uint32_t *__restrict__ counts, const uint32_t *__restrict__ ent_off,
             level_ref lv0, const uint32_t *__restrict__ map0, ge_rec *__restrict__ parts, uint32_t nblk, int ls0, seg_map map,
             uint32_t wstride, uint32_t blk0) {
    uint32_t t = (blk0 + blockIdx.x) * 256u + threadIdx.x;
    uint32_t e = 0, eend = 0;
    if (t < lv0.base[nblk]) {
        const uint32_t g = map0[t];
        const uint32_t rank = t - (lv0.local[g] + lv0.base[g >> 10]);
        const uint32_t cnt = counts[g];
        const uint32_t off = ent_off[g];
        // the bucket's k = ceil(cnt / S0) threads share its entries evenly (sizes differ by at most one): with fixed
        // S0-entry segments the last one is half empty on average and its lane idles while the wave runs S0 steps
        const uint32_t k = (cnt + (uint32_t)ls0 - 1) / (uint32_t)ls0;
        const uint32_t base = cnt / k, rem = cnt - base * k;
        const uint32_t lo = rank * base + (rank < rem ? rank : rem);
        e = off + lo;
        eend = e + base + (rank < rem ? 1u : 0u);
    }
    // Segments still differ in length (14, 15 or 16 entries under uniform digits) and a wave runs as long as its longest
    // lane: the block's 256 segments are dealt to its lanes in order of length (a counting sort in LDS), so that each of
    // its four waves gets segments of one length, or two.  Which lane sums a segment does not change the sum.
    if (ls0 < 256) {
        __shared__ uint32_t s_bin[256], s_wave[4], s_e[256], s_end[256], s_t[256];
        s_bin[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t len = eend - e;                     // 0 (no segment: past the end) .. S0
        const uint32_t mine = atomicAdd(&s_bin[len], 1u);
        __syncthreads();
        {                                                  // exclusive scan of the 256 bins: lane scan, then the waves' totals
            const uint32_t v = s_bin[threadIdx.x];
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            uint32_t x = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t y = __shfl_up(x, o, 64);
                if (lane >= o) x += y;
            }
            if (lane == 63) s_wave[wave] = x;
            __syncthreads();
            for (int w = 0; w < wave; w++) x += s_wave[w];
            s_bin[threadIdx.x] = x - v;
        }
        __syncthreads();
        const uint32_t pos = 255u - (s_bin[len] + mine);   // longest first: the empty segments end up in the last wave
        s_e[pos] = e;
        s_end[pos] = eend;
        s_t[pos] = t;
        __syncthreads();
        e = s_e[threadIdx.x];
        eend = s_end[threadIdx.x];
        t = s_t[threadIdx.x];
    }
    if (e >= eend) return;
    // variable-base: wstride = 0 and the entry is index | sign; fixed-base: window w's copy of point i
    // (2^(c w) P_i) lives wstride records further per window (<= 64 windows)
    auto record = [&](uint32_t en) {
        const uint32_t idx = wstride ? (en & FB_IDX_MASK) : (en & 0x7fffffffu);
        const uint32_t win = wstride ? ((en >> FB_WSHIFT) & FB_WMASK) : 0u;
        return record_of<MULTI>(map, idx) + (size_t)win * wstride;
    };
    ge acc;
    uint32_t en = entries[e];
    {
        ge_aniels_signed q;                                // identity + P = P
        load_aniels_signed(q, record(en), en >> 31);
        gez_from_aniels(acc, q);
    }
    for (e++; e < eend; e++) {
        en = entries[e];
        ge_aniels_signed q;
        load_aniels_signed(q, record(en), en >> 31);
        gez_madd_asm(acc, q);
    }
    store_ge(parts + t, acc);
}

// reduction pass `level` (>= 1): thread t merges up to S1 partials of level-1 of its bucket
// (map != nullptr: thread -> bucket map of this pass, see k_segmap; else binary search)
__global__ void __launch_bounds__(256) k_reduce(const uint32_t *__restrict__ counts, level_ref prev, level_ref cur,
                                                const uint32_t *__restrict__ map, const ge_rec *__restrict__ parts_in,
                                                ge_rec *__restrict__ parts_out, uint32_t nblk, uint32_t ne, int level, int ls0,
                                                int ls1) {
    uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= cur.base[nblk]) return;
    uint32_t rank, g;
    if (map) {
        g = map[t];
        rank = t - (cur.local[g] + cur.base[g >> 10]);
    } else {
        g = find_bucket(cur, t, nblk, ne, rank);
    }
    uint32_t kprev = seg_count(counts[g], level - 1, ls0, ls1);
    uint32_t in0 = prev.local[g] + prev.base[g >> 10];
    uint32_t lo = rank << ls1, hi = lo + (1u << ls1);
    hi = hi < kprev ? hi : kprev;
    fe_zeros Zr;
    fe_zeros_init(Zr);
    ge acc;
    load_ge(acc, parts_in + in0 + lo);
    for (uint32_t i = lo + 1; i < hi; i++) {
        ge b;
        load_ge(b, parts_in + in0 + i);
        gez_add(acc, acc, b, Zr);
    }
    store_ge(parts_out + t, acc);
}

// The same merge for small problems (a proof-sized MSM has ~16 k buckets: one thread per merge
// leaves most of the GPU idle while each walks up to 15 dependent additions).  G lanes per merge:
// every lane adds 16/G partials, log2(G) butterfly steps add the lanes up (S1 must be 16).
// A little more total work, a fraction of the latency; chosen by the host when buckets are few.
template <int G>
__global__ void __launch_bounds__(256) k_reduce_wide(const uint32_t *__restrict__ counts, level_ref prev, level_ref cur,
                                                     const uint32_t *__restrict__ map, const ge_rec *__restrict__ parts_in,
                                                     ge_rec *__restrict__ parts_out, uint32_t nblk, uint32_t ne, int level,
                                                     int ls0) {
    constexpr int PER = 16 / G;
    const uint32_t lane_id = blockIdx.x * 256u + threadIdx.x;
    const uint32_t t = lane_id / G, sub = lane_id % G;
    if (t >= cur.base[nblk]) return;               // uniform over the G lanes of a merge
    uint32_t rank, g;
    if (map) {
        g = map[t];
        rank = t - (cur.local[g] + cur.base[g >> 10]);
    } else {
        g = find_bucket(cur, t, nblk, ne, rank);
    }
    const uint32_t kprev = seg_count(counts[g], level - 1, ls0, 4);
    const uint32_t in0 = prev.local[g] + prev.base[g >> 10];
    const uint32_t first = (rank << 4) + sub * PER;
    ge acc;
    if (first < kprev) load_ge(acc, parts_in + in0 + first);
    else ge_identity(acc);
    for (int k = 1; k < PER; k++) {
        if (first + k < kprev) {
            ge b;
            load_ge(b, parts_in + in0 + first + k);
            ge_add(acc, acc, b);
        }
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) {
        ge other;
        ge_shfl_xor(other, acc, o, G);
        ge_add(acc, acc, other);
    }
    if (sub == 0) store_ge(parts_out + t, acc);
}

struct levels_dev {
    level_ref lv[MAX_LEVELS];
    const ge_rec *parts[MAX_LEVELS];
    int nlevels;          // levels 0 .. nlevels-1 exist
};

// sum of bucket g: lives at the first level whose segment count is 1
__device__ __forceinline__ void bucket_sum(ge &acc, const levels_dev &L, const uint32_t *counts, size_t g, int ls0, int ls1) {
    uint32_t cnt = counts[g];
    if (cnt == 0) {
        ge_identity(acc);
        return;
    }
    uint32_t k = (cnt + (uint32_t)ls0 - 1) / (uint32_t)ls0;
    int r = 0;
    while (k > 1 && r + 1 < L.nlevels) {
        k = (k + (1u << ls1) - 1) >> ls1;
        r++;
    }
    load_ge(acc, L.parts[r] + L.lv[r].local[g] + L.lv[r].base[g >> 10]);
}

// tail A: one block per chunk of `chunk` buckets of one (target, window): bit tree
// over the bucket index; writes log2(chunk)+1 points per chunk.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) k_tail_a(levels_dev L, const uint32_t *__restrict__ counts, ge_rec *__restrict__ out,
                                                int NB, int chunk, int logchunk, int W, int ls0, int ls1) {
    extern __shared__ uint32_t lds[];
    const int cidx = blockIdx.x, w = blockIdx.y, t = blockIdx.z;
    const size_t win = (size_t)t * W + w;
    const size_t g = win * NB + (size_t)cidx * chunk + threadIdx.x;
    ge acc;
    bucket_sum(acc, L, counts, g, ls0, ls1);
    lds_put(lds, chunk, threadIdx.x, acc);
    __syncthreads();
    lds_bit_tree(lds, chunk, chunk, 1, logchunk);
    const int per = logchunk + 1;
    if ((int)threadIdx.x < per) {
        ge o;
        lds_get(o, lds, chunk, threadIdx.x);
        store_ge(out + (win * gridDim.x + cidx) * per + threadIdx.x, o);
    }
}

// tail B: one block per (target, window): continue the tree across chunks.
__global__ void __launch_bounds__(320) k_tail_b(const ge_rec *__restrict__ chunks, ge_rec *__restrict__ out, int nchunks,
                                                int lognchunks, int per0, int W) {
    extern __shared__ uint32_t lds[];
    const int w = blockIdx.x, t = blockIdx.y;
    const size_t win = (size_t)t * W + w;
    const int nslots = nchunks * per0;
    if ((int)threadIdx.x < nslots) {
        ge a;
        load_ge(a, chunks + win * nslots + threadIdx.x);
        lds_put(lds, nslots, threadIdx.x, a);
    }
    __syncthreads();
    lds_bit_tree(lds, nslots, nchunks, per0, lognchunks);
    const int per = per0 + lognchunks;
    if ((int)threadIdx.x < per) {
        ge o;
        lds_get(o, lds, nslots, threadIdx.x);
        store_ge(out + win * per + threadIdx.x, o);
    }
}

// ------------------------------------------------------------------ folded generators
// G'_i = sum_{j = i mod m} s_j G_j (and H' likewise): 2m sums per scalar row [s_G (np) | s_H (np)], each over
// np / m generators with full-size scalars, from the fixed-base table (window w of point j is one record).
// Two bucket layouts (fold_shape):
//  * narrow digits (c <= 7, read from a table that holds a block per c bits -- the generator table's half windows): a
//    signed digit |d| <= 2^(c-1) goes, with its sign, into bucket (target, |d| - 1): ONE addition per digit, 2^(c-1)
//    buckets per target.  At c = 7 a target's 32 generators x 36 windows cost 1152 additions + 128 for the set.
//  * wide digits (c <= 14, the table's own windows): |d| = p0 + 32 p1 + 1024 p2 is cut into 5-bit pieces and the window's
//    point goes into bucket (target, piece t, p_t): 3 x 32 buckets per target and THREE additions per digit
//    (32 x 18 x 3 = 1728 + 192 at c = 14).
// The sort, accumulate and merge kernels above then run on those bucket numbers, k_fold_sets turns every bucket set
// into sum_b (b+1) B_b, and k_fold_finish joins a target's sets (x 32, x 1024) into one point and writes it as an
// affine Niels record, ready to be a point of a later MSM.
struct fold_shape {
    int P, PB;                                             // pieces per digit, bits per piece: 2^PB buckets per set
};
__device__ __forceinline__ uint32_t fold_target(uint32_t row, uint32_t i, uint32_t np, uint32_t m) {
    const uint32_t side = i >= np ? 1u : 0u, jj = i - side * np;
    return (row * 2u + side) * m + (jj & (m - 1u));
}
// bucket of piece t (value pc in [1, 2^PB]; a single piece uses all 2^PB slots, 5-bit pieces never reach slot 31)
__device__ __forceinline__ uint32_t fold_piece(uint32_t a, int t, fold_shape fs) {
    return fs.P == 1 ? a : (a >> (fs.PB * t)) & ((1u << fs.PB) - 1u);
}
// bins of 256 buckets; one tile (a run of indices of one row, one side, one window) reaches the
// m * P * 2^PB / 256 bins of its (row, side), which is what the LDS cursors cover
__global__ void __launch_bounds__(512) k_fold_hist(const int16_t *__restrict__ digits, uint32_t *__restrict__ bin_counts, size_t n,
                                                   int W, uint32_t np, uint32_t m, int tile, fold_shape fs) {
    extern __shared__ uint32_t lds[];
    const int w = blockIdx.y, row = blockIdx.z;
    const int16_t *dg = digits + ((size_t)row * W + w) * n;
    const uint32_t start = blockIdx.x * (uint32_t)tile, end = start + tile < n ? start + tile : (uint32_t)n;
    const uint32_t nb = ((m * (uint32_t)fs.P) << fs.PB) >> 8;                      // bins per (row, side)
    const uint32_t binbase = ((uint32_t)row * 2u + (start >= np ? 1u : 0u)) * nb;
    for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) lds[b] = 0;
    __syncthreads();
    for (uint32_t i0 = start; i0 < end; i0 += blockDim.x) {
        const uint32_t i = i0 + threadIdx.x;
        const int d = i < end ? dg[i] : 0;
        const uint32_t a = (uint32_t)(d < 0 ? -d : d);
        const uint32_t tg = fold_target(row, i < end ? i : start, np, m);
        for (int t = 0; t < fs.P; t++) {
            const uint32_t pc = fold_piece(a, t, fs);
            const uint32_t g = ((tg * (uint32_t)fs.P + t) << fs.PB) + pc - 1u;
            (void)lds_claim(lds, (g >> 8) - binbase, pc != 0);
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) {
        const uint32_t c = lds[b];
        if (c) atomicAdd(&bin_counts[binbase + b], c);
    }
}
__global__ void __launch_bounds__(512) k_fold_scatter(const int16_t *__restrict__ digits, const uint32_t *__restrict__ bin_off,
                                                      uint32_t *__restrict__ bin_fill, uint2 *__restrict__ tmp, size_t n, int W,
                                                      uint32_t np, uint32_t m, int tile, fold_shape fs) {
    extern __shared__ uint32_t lds[];
    const int w = blockIdx.y, row = blockIdx.z;
    const int16_t *dg = digits + ((size_t)row * W + w) * n;
    const uint32_t start = blockIdx.x * (uint32_t)tile, end = start + tile < n ? start + tile : (uint32_t)n;
    const uint32_t nb = ((m * (uint32_t)fs.P) << fs.PB) >> 8;
    const uint32_t binbase = ((uint32_t)row * 2u + (start >= np ? 1u : 0u)) * nb;
    for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) lds[b] = 0;
    __syncthreads();
    for (uint32_t i0 = start; i0 < end; i0 += blockDim.x) {
        const uint32_t i = i0 + threadIdx.x;
        const int d = i < end ? dg[i] : 0;
        const uint32_t a = (uint32_t)(d < 0 ? -d : d);
        const uint32_t tg = fold_target(row, i < end ? i : start, np, m);
        for (int t = 0; t < fs.P; t++) {
            const uint32_t pc = fold_piece(a, t, fs);
            const uint32_t g = ((tg * (uint32_t)fs.P + t) << fs.PB) + pc - 1u;
            (void)lds_claim(lds, (g >> 8) - binbase, pc != 0);
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) {
        const uint32_t c = lds[b];
        if (c) lds[b] = bin_off[binbase + b] + atomicAdd(&bin_fill[binbase + b], c);
    }
    __syncthreads();
    for (uint32_t i0 = start; i0 < end; i0 += blockDim.x) {
        const uint32_t i = i0 + threadIdx.x;
        const int d = i < end ? dg[i] : 0;
        const uint32_t a = (uint32_t)(d < 0 ? -d : d);
        const uint32_t tg = fold_target(row, i < end ? i : start, np, m);
        const uint32_t entry = i | ((uint32_t)w << FB_WSHIFT) | (d < 0 ? 0x80000000u : 0u);
        for (int t = 0; t < fs.P; t++) {
            const uint32_t pc = fold_piece(a, t, fs);
            const uint32_t g = ((tg * (uint32_t)fs.P + t) << fs.PB) + pc - 1u;
            const uint32_t pos = lds_claim(lds, (g >> 8) - binbase, pc != 0);
            if (pc != 0) tmp[pos] = make_uint2(entry, g & 255u);
        }
    }
}
// one thread per bucket set: sum_b (b + 1) B_b over its 2^PB buckets by running sums (a 32-thread bit tree per set,
// k_tail_a's way, took 1.9 ms for the 98 k sets of eight 2^16-multiplier proofs; this takes a tenth)
__global__ void __launch_bounds__(256) k_fold_sets(levels_dev L, const uint32_t *__restrict__ counts, uint32_t nsets,
                                                  ge_rec *__restrict__ out, int ls0, int ls1, int PB) {
    const uint32_t st = blockIdx.x * 256u + threadIdx.x;
    if (st >= nsets) return;
    ge run, acc, b;
    ge_identity(run);
    ge_identity(acc);
    for (int k = (1 << PB) - 1; k >= 0; k--) {
        const size_t g = ((size_t)st << PB) + k;
        if (counts[g]) {
            bucket_sum(b, L, counts, g, ls0, ls1);
            ge_add(run, run, b);
        }
        ge_add(acc, acc, run);
    }
    store_ge(out + st, acc);
}
// sets: [target][piece] as k_fold_sets leaves them; one thread per target
__global__ void __launch_bounds__(256) k_fold_finish(const ge_rec *__restrict__ sets, uint32_t ntargets, aniels_rec *__restrict__ tab,
                                                    fold_shape fs) {
    const uint32_t tg = blockIdx.x * 256u + threadIdx.x;
    if (tg >= ntargets) return;
    ge p;
    for (int t = fs.P - 1; t >= 0; t--) {
        ge v;
        load_ge(v, sets + (size_t)tg * fs.P + t);
        if (t == fs.P - 1) {
            p = v;
        } else {
            for (int k = 0; k < fs.PB; k++) ge_double(p, p);
            ge_add(p, p, v);
        }
    }
    fe zinv;
    fe_invert(zinv, p.Z);
    ge_aniels r;
    ge_to_aniels(r, p, zinv);
    uint4 *o = reinterpret_cast<uint4 *>(tab + tg);
    o[0] = make_uint4(r.hYpX.v[0], r.hYpX.v[1], r.hYpX.v[2], r.hYpX.v[3]);
    o[1] = make_uint4(r.hYpX.v[4], r.hYpX.v[5], r.hYpX.v[6], r.hYpX.v[7]);
    o[2] = make_uint4(r.hYmX.v[0], r.hYmX.v[1], r.hYmX.v[2], r.hYmX.v[3]);
    o[3] = make_uint4(r.hYmX.v[4], r.hYmX.v[5], r.hYmX.v[6], r.hYmX.v[7]);
    o[4] = make_uint4(r.Tdxy.v[0], r.Tdxy.v[1], r.Tdxy.v[2], r.Tdxy.v[3]);
    o[5] = make_uint4(r.Tdxy.v[4], r.Tdxy.v[5], r.Tdxy.v[6], r.Tdxy.v[7]);
    fe nt;
    fe_neg(nt, r.Tdxy);
    o[6] = make_uint4(nt.v[0], nt.v[1], nt.v[2], nt.v[3]);
    o[7] = make_uint4(nt.v[4], nt.v[5], nt.v[6], nt.v[7]);
}

// ------------------------------------------------------------------ host side

static int ilog2(size_t x) {
    int r = 0;
    while ((size_t)1 << (r + 1) <= x) r++;
    return r;
}

// variable-base window size: the candidate that minimises windows * (n + 18 * buckets).  The weight of a bucket is
// measured, not counted (a bucket costs ~2 full additions in the merge and tail kernels, which run far below the
// accumulate kernel's rate): at n = 2^20, c = 16 takes 1.234 ms per MSM against 1.010 ms for c = 14 although its
// accumulate pass is shorter; at 2^22 the two tie (4.16 ms); beyond that c = 16 wins.  (profiles/r02_window_sweep.txt)
static int auto_window(size_t n) {
    static const int cand[] = {4, 6, 7, 9, 12, 14, 16};
    int best = 4;
    double best_cost = 1e300;
    for (int c : cand) {
        double cost = (double)windows_for(c) * ((double)n + 18.0 * (double)(1u << (c - 1)));
        if (cost < best_cost) {
            best_cost = cost;
            best = c;
        }
    }
    return best;
}

// fixed-base window size for a generator table of `points` points (shared bucket set:
// bucket overhead is paid once, not per window)
// (at most 32 windows: the entry word keeps the window in 5 bits)
int fixed_base_window(size_t points) {
    if (points <= ((size_t)1 << 12)) return 9;
    if (points <= ((size_t)1 << 15)) return 12;
    return 14;
}
int fixed_base_windows(int c) { return windows_for(c); }

msm_plan make_plan(size_t n, int T, const msm_tuning &tune, int fixed_c, float density, int resident_lanes) {
    msm_plan pl{};
    // density: the fraction of a target's scalars that are non-zero (block-diagonal rows over a union of tables)
    size_t n_eff = n;
    if (!fixed_c && density > 0.f && density < 1.f) n_eff = (size_t)((double)n * density) + 1;
    int c = auto_window(n_eff);
    if (tune.c >= 2 && tune.c <= 16) c = tune.c;
    if (fixed_c) c = fixed_c;
    pl.c = c;
    pl.W = windows_for(c);
    pl.Wb = fixed_c ? 1 : pl.W;
    pl.NB = 1 << (c - 1);
    // S0 = entries per accumulate thread (any integer).  32 since the pass deals a block's segments to its lanes by
    // length (k_accumulate): half the partial sums of S0 = 16, so half the merge pass, and the longer segments no
    // longer cost the pass lanes that idle behind a wave's longest one.  2^20 points, four in flight: 1.03-1.05 ms per
    // MSM at 16, 0.965-0.98 at 32-56; 2^16 ... 2^22: equal or better at every size but 2^18 (-3 %); proofs/s
    // unchanged (profiles/r03_msm_segment_sweep.txt).  Before that sort, 16 ... 64 were all within 1 %.
    // Shrinking S0 so that a proof-sized MSM's threads just fill the machine does not pay either.
    (void)resident_lanes;
    pl.ls0 = (tune.ls0 >= 1 && tune.ls0 <= 256) ? tune.ls0 : 32;
    pl.ls1 = (tune.ls1 >= 1 && tune.ls1 <= 8) ? tune.ls1 : 4;
    pl.P = pl.ls0;
    // passes: S0 * S1^R >= the most entries one bucket can hold
    const size_t worst = fixed_c ? n * (size_t)pl.W : n;
    int R = 0;
    while (((size_t)pl.ls0 << (pl.ls1 * R)) < worst) R++;
    pl.R = R;
    int tile = 4096;
    while (tile < 32768 && (size_t)tile * 64 < n) tile *= 2;
    if (tune.tile >= 512 && tune.tile <= (1 << 20)) tile = tune.tile;
    pl.tile = tile;
    pl.ntiles = (int)((n + tile - 1) / tile);
    if (pl.ntiles < 1) pl.ntiles = 1;
    pl.n = n;
    pl.T = T;
    // bins of the two-level sort: ~8k entries each, at most 256 buckets per bin
    {
        const size_t items = (fixed_c ? (size_t)pl.W : (size_t)1) * n;
        int nbin = 1;
        while (nbin < pl.NB && items / (size_t)nbin > 8192) nbin <<= 1;
        while (pl.NB / nbin > 256) nbin <<= 1;
        pl.nbin = nbin;
        pl.lowshift = 0;
        while ((nbin << pl.lowshift) < pl.NB) pl.lowshift++;
    }
    pl.sort_chunk = (tune.sort_chunk >= 256 && tune.sort_chunk <= (1 << 20)) ? tune.sort_chunk : 12288;
    pl.chunk = pl.NB < 256 ? pl.NB : 256;
    pl.nchunks = pl.NB / pl.chunk;
    pl.outs = c;
    return pl;
}

// bpg_debug_fail_alloc(k): the k-th device allocation from now on (1 = the next one) fails as if the GPU were
// out of memory; 0 disarms.  For the tests of the failure paths (no 288 GB allocation needed to get there).
static std::atomic<int> g_fail_alloc{0};
void debug_fail_alloc(int countdown) { g_fail_alloc.store(countdown); }
hipError_t device_malloc(void **p, size_t bytes) {
    int c = g_fail_alloc.load();
    while (c > 0 && !g_fail_alloc.compare_exchange_weak(c, c - 1)) {}
    if (c == 1) {
        *p = nullptr;
        return hipErrorOutOfMemory;
    }
    // fail at once when the device cannot hold the request (workspaces grow to GBs per prover thread: 64 threads in
    // groups of sixteen ask for more than the 288 GB): asking the driver anyway was seen to stall for minutes
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && bytes > free_b) {
        *p = nullptr;
        return hipErrorOutOfMemory;
    }
    return hipMalloc(p, bytes);
}

int grow(void **p, size_t *cap, size_t need) {
    if (need <= *cap) return BPG_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    size_t want = need + need / 8 + 256;
    BPG_CHECK_HIP(device_malloc(p, want));
    *cap = want;
    return BPG_OK;
}

int grow_pinned(void **p, size_t *cap, size_t need) {
    if (need <= *cap) return BPG_OK;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr;
    *cap = 0;
    size_t want = need + need / 8 + 4096;
    // no flags = COHERENT host memory in HIP's rules (HIP_HOST_COHERENT is ignored then): kernels that write results
    // straight into these buffers (MSM tails, k_pedersen, k_small_copies) rely on it -- their stores are not cached on
    // the device and are visible to the host once the stream's event has completed
    if (hipHostMalloc(p, want, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        *p = nullptr;
        return BPG_ENOMEM;
    }
    *cap = want;
    return BPG_OK;
}

template <int C>
static void launch_digits(hipStream_t s, const uint32_t *sc, int16_t *dg, size_t n, size_t total, const sc_map &sm,
                          uint32_t *zero, uint32_t nzero) {
    unsigned blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(k_digits<C>, dim3(blocks), dim3(256), 0, s, sc, dg, n, total, sm, zero, nzero);
}

static void dispatch_digits(int c, hipStream_t s, const uint32_t *sc, int16_t *dg, size_t n, size_t total, const sc_map &sm,
                            uint32_t *zero, uint32_t nzero) {
    switch (c) {
#define BPG_CASE(C) case C: launch_digits<C>(s, sc, dg, n, total, sm, zero, nzero); break;
        BPG_CASE(2) BPG_CASE(3) BPG_CASE(4) BPG_CASE(5) BPG_CASE(6) BPG_CASE(7) BPG_CASE(8) BPG_CASE(9)
        BPG_CASE(10) BPG_CASE(11) BPG_CASE(12) BPG_CASE(13) BPG_CASE(14) BPG_CASE(15) BPG_CASE(16)
#undef BPG_CASE
    }
}

static bpg_msm_job *acquire_job(bpg_lane *lane) {
    for (auto *j : lane->jobs)
        if (!j->busy.load(std::memory_order_acquire)) return j;
    auto *j = new bpg_msm_job();
    j->lane = lane;
    if (hipStreamCreateWithFlags(&j->stream, hipStreamNonBlocking) != hipSuccess) {
        delete j;
        return nullptr;
    }
    for (auto &e : j->ev)
        if (hipEventCreate(&e) != hipSuccess) {
            msm_job_destroy(j);                     // frees the stream and the events created so far
            return nullptr;
        }
    lane->jobs.push_back(j);
    return j;
}

void msm_job_destroy(bpg_msm_job *j) {
    if (!j) return;
    (void)hipStreamSynchronize(j->stream);
    (void)hipFree(j->d_digits);
    (void)hipFree(j->d_entries);
    (void)hipFree(j->d_counts);
    (void)hipFree(j->d_partials);
    (void)hipFree(j->d_chunk);
    (void)hipFree(j->d_out);
    (void)hipFree(j->d_stage);
    (void)hipFree(j->d_tmp);
    if (j->h_out) (void)hipHostFree(j->h_out);
    for (auto &e : j->ev)
        if (e) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(j->stream);
    delete j;
}

// ---- one Pippenger run: geometry, workspace, and the three groups of launches.  msm_submit() takes the workspace
//      from a job's growable buffers; msm_noalloc() (further down) from the library's static device array.
struct msm_geom {
    size_t NE, M, nblk, lvl_words, map_words, NBT, NCH, part_total;
    size_t cap_lv[MAX_LEVELS], part_off[MAX_LEVELS];
    int nlevels, logchunk, lognch, per0, LOWN, ngroups;
    size_t chunk2_off;
    uint32_t CH;
    // bytes of each workspace buffer
    size_t b_digits, b_entries, b_counts, b_tmp, b_partials, b_chunk, b_out;
};
struct msm_bufs {
    int16_t *digits;
    uint32_t *entries, *counts;
    void *tmp;                 // level-1 items: uint2 each, or one packed word each (sort_item)
    ge_rec *partials, *chunk, *out;
};
struct msm_maps {
    seg_map map;
    sc_map smap;
};

static int msm_geometry(const msm_plan &pl, size_t n, int T, msm_geom &g) {
    g.NE = (size_t)T * pl.Wb * pl.NB;                      // buckets
    g.M = (size_t)T * pl.W * n;                            // upper bound on entries
    if (g.M >= ((size_t)1 << 32) || g.NE >= ((size_t)1 << 30)) return BPG_EINVAL;
    g.nblk = (g.NE + 1023) / 1024;
    g.logchunk = ilog2(pl.chunk);
    g.lognch = ilog2(pl.nchunks);
    g.per0 = g.logchunk + 1;
    g.nlevels = pl.R + 1;                                  // levels 0..R
    if (g.nlevels > MAX_LEVELS) return BPG_EINVAL;
    // partial-sum capacity per level: level 0 <= M/S0 + NE, each further level at most half
    g.cap_lv[0] = g.M / (size_t)pl.ls0 + g.NE + 1;
    for (int r = 1; r < g.nlevels; r++) {
        // a bucket with cnt entries has <= cnt / (S0 S1^r) + 1 threads at level r, and at most half of level r-1
        const size_t by_count = ((g.M / (size_t)pl.ls0) >> (r * pl.ls1)) + g.NE + 1;
        g.cap_lv[r] = g.cap_lv[r - 1] / 2 + 1 < by_count ? g.cap_lv[r - 1] / 2 + 1 : by_count;
    }
    g.part_total = 0;
    for (int r = 0; r < g.nlevels; r++) {
        g.part_off[r] = g.part_total;
        g.part_total += g.cap_lv[r];
    }
    g.lvl_words = g.NE + g.nblk + 1;                       // local[NE] | base[nblk + 1]
    g.map_words = g.cap_lv[0] + (g.nlevels > 1 ? g.cap_lv[1] : 0);
    g.NBT = (size_t)T * pl.Wb * pl.nbin;                   // bins
    g.CH = (uint32_t)pl.sort_chunk;                        // items per level-2 block
    g.LOWN = pl.NB / pl.nbin;                              // buckets per bin
    g.NCH = g.M / g.CH + g.NBT;                            // upper bound on chunks: sum of max(1, ceil(cnt / CH))
    g.b_digits = g.M * sizeof(int16_t);
    g.b_entries = g.M * sizeof(uint32_t);
    g.b_counts = (2 * g.NE + (size_t)(g.nlevels + 1) * g.lvl_words + g.map_words + 4 * g.NBT + g.NCH * (size_t)g.LOWN + 16) * sizeof(uint32_t);
    g.b_tmp = g.M * sizeof(uint2) + 64;                    // level-1 output: (entry, bucket in bin)
    g.b_partials = g.part_total * sizeof(ge_rec);
    // tail: chunks of 256 buckets -> (more than 32 chunks per window: groups of 32 chunks ->) one set of c points per window
    g.ngroups = pl.nchunks > 32 ? pl.nchunks / 32 : 0;
    g.chunk2_off = (size_t)T * pl.Wb * pl.nchunks * g.per0;
    g.b_chunk = (g.chunk2_off + (size_t)T * pl.Wb * g.ngroups * (g.per0 + 5)) * sizeof(ge_rec);
    g.b_out = (size_t)T * pl.Wb * pl.outs * sizeof(ge_rec);
    return BPG_OK;
}

static int msm_make_maps(const msm_seg *segs, int nseg, size_t n, size_t sc_stride, msm_maps &mm) {
    size_t acc = 0;
    for (int k = 0; k < 4; k++) {
        mm.smap.off[k] = 0;
        if (k < nseg) {
            if (segs[k].len && !segs[k].base) return BPG_EINVAL;
            if (sc_stride && (segs[k].sc_off + segs[k].len > sc_stride || sc_stride >= ((size_t)1 << 32))) return BPG_EINVAL;
            mm.map.base[k] = segs[k].base - acc;
            mm.smap.off[k] = (uint32_t)(sc_stride ? segs[k].sc_off : acc);
            acc += segs[k].len;
        } else {
            mm.map.base[k] = nullptr;
        }
        mm.map.len[k] = k < nseg ? (uint32_t)acc : 0xffffffffu;
        mm.smap.end[k] = mm.map.len[k];
    }
    if (acc != n) return BPG_EINVAL;
    mm.smap.stride = sc_stride ? sc_stride : n;
    return BPG_OK;
}

// where things live inside the counts buffer
struct msm_layout {
    uint32_t *counts, *ent_off, *lvl_base, *map0, *map1, *bin_counts, *bin_fill, *bin_off, *chunk_off, *chunk_hist;
    levels_dev L;
};
static msm_layout msm_lay_out(const msm_geom &g, const msm_bufs &b) {
    msm_layout y{};
    y.counts = b.counts;
    y.ent_off = y.counts + g.NE;
    y.lvl_base = y.ent_off + g.NE;                         // level -1 (unused), then levels 0..R
    y.map0 = y.lvl_base + (size_t)(g.nlevels + 1) * g.lvl_words;
    y.map1 = g.nlevels > 1 ? y.map0 + g.cap_lv[0] : nullptr;
    y.bin_counts = y.map0 + g.map_words;
    y.bin_fill = y.bin_counts + g.NBT;
    y.bin_off = y.bin_fill + g.NBT;
    y.chunk_off = y.bin_off + g.NBT + 1;
    y.chunk_hist = y.chunk_off + g.NBT + 1;
    y.L.nlevels = g.nlevels;
    for (int r = 0; r < g.nlevels; r++) {
        y.L.lv[r] = level_ref{y.lvl_base + (size_t)(r + 1) * g.lvl_words, y.lvl_base + (size_t)(r + 1) * g.lvl_words + g.NE};
        y.L.parts[r] = b.partials + g.part_off[r];
    }
    return y;
}

// digits, two-level bucket sort, thread offsets and thread -> bucket maps of every pass
static void msm_enqueue_sort(hipStream_t s, const msm_plan &pl, const msm_geom &g, const msm_bufs &b, const msm_layout &y,
                             const uint32_t *d_sc, size_t n, int T, const sc_map &smap, int shared) {
    dispatch_digits(pl.c, s, d_sc, b.digits, n, (size_t)T * n, smap, y.bin_counts, (uint32_t)(2 * g.NBT));
    dim3 tgrid(pl.ntiles, pl.W, T);
    const size_t bin_lds = (size_t)pl.nbin * sizeof(uint32_t);
    hipLaunchKernelGGL(k_bin_hist, tgrid, dim3(512), bin_lds, s, b.digits, y.bin_counts, n, pl.W, pl.nbin, pl.lowshift, pl.tile, shared);
    hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, s, y.bin_counts, y.bin_off, y.chunk_off, (uint32_t)g.NBT, g.CH);
    const bool packed = !shared && n <= SORT_PACK_MAX_POINTS && g.LOWN <= 256;
    const dim3 place_grid((unsigned)(g.NCH < 2048 ? g.NCH : 2048));
    if (packed) {
        hipLaunchKernelGGL(k_bin_scatter<true>, tgrid, dim3(512), bin_lds, s, b.digits, y.bin_off, y.bin_fill, (uint32_t *)b.tmp, n, pl.W,
                           pl.nbin, pl.lowshift, pl.tile, shared);
        hipLaunchKernelGGL(k_chunk_count<true>, dim3((unsigned)g.NCH), dim3(256), 0, s, y.bin_off, y.chunk_off, (const uint32_t *)b.tmp,
                           y.chunk_hist, y.counts, y.ent_off, b.entries, (uint32_t)g.NBT, g.CH, g.LOWN);
    } else {
        hipLaunchKernelGGL(k_bin_scatter<false>, tgrid, dim3(512), bin_lds, s, b.digits, y.bin_off, y.bin_fill, (uint2 *)b.tmp, n, pl.W,
                           pl.nbin, pl.lowshift, pl.tile, shared);
        hipLaunchKernelGGL(k_chunk_count<false>, dim3((unsigned)g.NCH), dim3(256), 0, s, y.bin_off, y.chunk_off, (const uint2 *)b.tmp,
                           y.chunk_hist, y.counts, y.ent_off, b.entries, (uint32_t)g.NBT, g.CH, g.LOWN);
    }
    // (2b and 2c have work only when some bin is larger than a chunk: modest grids that stride)
    hipLaunchKernelGGL(k_chunk_offsets, dim3((unsigned)(g.NBT < 512 ? g.NBT : 512)), dim3(256), 0, s, y.bin_off, y.chunk_off,
                       y.chunk_hist, y.counts, y.ent_off, (uint32_t)g.NBT, g.LOWN);
    if (packed)
        hipLaunchKernelGGL(k_chunk_place<true>, place_grid, dim3(256), 0, s, y.bin_off, y.chunk_off, (const uint32_t *)b.tmp,
                           y.chunk_hist, y.ent_off, b.entries, (uint32_t)g.NBT, g.CH, g.LOWN);
    else
        hipLaunchKernelGGL(k_chunk_place<false>, place_grid, dim3(256), 0, s, y.bin_off, y.chunk_off, (const uint2 *)b.tmp,
                           y.chunk_hist, y.ent_off, b.entries, (uint32_t)g.NBT, g.CH, g.LOWN);
    // thread offsets of every pass (levels 0..R), two launches
    hipLaunchKernelGGL(k_scan_a, dim3((unsigned)g.nblk), dim3(256), 0, s, y.counts, y.lvl_base, g.lvl_words, g.NE, g.nlevels + 1,
                       pl.ls0, pl.ls1);
    hipLaunchKernelGGL(k_scan_b, dim3(g.nlevels + 1), dim3(1024), 0, s, y.lvl_base, g.lvl_words, g.NE, g.nblk);
    // thread -> bucket maps of levels 0 and 1 (level 1 only exists when nlevels > 1)
    hipLaunchKernelGGL(k_segmap, dim3((unsigned)((g.NE + 255) / 256)), dim3(256), 0, s, y.counts, y.L.lv[0],
                       y.L.lv[g.nlevels > 1 ? 1 : 0], y.map0, y.map1, (uint32_t)g.NE, pl.ls0, pl.ls1);
}

// ev_start / ev_stop (may be null): recorded at the dispatch's own begin and end, the timestamps a kernel trace shows
// (events recorded around the launch instead also time the barrier packets and the cache write-back between them)
static void msm_enqueue_accumulate(hipStream_t s, const msm_plan &pl, const msm_geom &g, const msm_bufs &b, const msm_layout &y,
                                   const seg_map &map, bool multi, uint32_t wstride, hipEvent_t ev_start = nullptr,
                                   hipEvent_t ev_stop = nullptr) {
    const unsigned blocks = (unsigned)((g.cap_lv[0] + 255) / 256);
    static const size_t acc_lds = [] {
        const char *e = getenv("BPG_ACC_LDS");
        return e ? (size_t)atoi(e) : (size_t)0;
    }();
    // BPG_ACC_CHUNK=k: the pass goes out in launches of at most k blocks (experiment: streams share the hardware
    // queues and a queue runs one kernel at a time, so a small kernel of another thread's round waits for a whole pass
    // in front of it; measured, chunking changes nothing -- the queues idle between kernels, not inside them).  Off.
    static const unsigned chunk = [] {
        const char *e = getenv("BPG_ACC_CHUNK");
        const int v = e ? atoi(e) : 0;
        return (unsigned)(v > 0 ? v : 0x7fffffff);
    }();
    for (unsigned b0 = 0; b0 < blocks || b0 == 0; b0 += chunk) {
        const unsigned nb = blocks - b0 < chunk ? blocks - b0 : chunk;
        const dim3 grid(nb ? nb : 1);
        hipEvent_t e0 = b0 == 0 ? ev_start : nullptr, e1 = b0 + nb >= blocks ? ev_stop : nullptr;
        if (multi)
            hipExtLaunchKernelGGL((k_accumulate<true>), grid, dim3(256), acc_lds, s, e0, e1, 0, b.entries, y.counts, y.ent_off,
                                  y.L.lv[0], y.map0, b.partials + g.part_off[0], (uint32_t)g.nblk, pl.ls0, map, wstride, (uint32_t)b0);
        else
            hipExtLaunchKernelGGL((k_accumulate<false>), grid, dim3(256), acc_lds, s, e0, e1, 0, b.entries, y.counts, y.ent_off,
                                  y.L.lv[0], y.map0, b.partials + g.part_off[0], (uint32_t)g.nblk, pl.ls0, map, wstride, (uint32_t)b0);
        if (!nb) break;
    }
}

// merge passes and the bit-tree tail: leaves pl.outs points per bucket set in b.out
static void msm_enqueue_merge_tail(hipStream_t s, const msm_plan &pl, const msm_geom &g, const msm_bufs &b, const msm_layout &y, int T) {
    const bool wide = pl.ls1 == 4 && g.NE <= 32768;        // latency-bound regime: 4 lanes per merge
    for (int r = 1; r < g.nlevels; r++) {
        const uint32_t *mp = r == 1 ? (const uint32_t *)y.map1 : (const uint32_t *)nullptr;
        if (wide)
            hipLaunchKernelGGL((k_reduce_wide<4>), dim3((unsigned)((g.cap_lv[r] * 4 + 255) / 256)), dim3(256), 0, s, y.counts,
                               y.L.lv[r - 1], y.L.lv[r], mp, b.partials + g.part_off[r - 1], b.partials + g.part_off[r],
                               (uint32_t)g.nblk, (uint32_t)g.NE, r, pl.ls0);
        else
            hipLaunchKernelGGL(k_reduce, dim3((unsigned)((g.cap_lv[r] + 255) / 256)), dim3(256), 0, s, y.counts, y.L.lv[r - 1],
                               y.L.lv[r], mp, b.partials + g.part_off[r - 1], b.partials + g.part_off[r], (uint32_t)g.nblk,
                               (uint32_t)g.NE, r, pl.ls0, pl.ls1);
    }
    ge_rec *a_out = pl.nchunks == 1 ? b.out : b.chunk;
    hipLaunchKernelGGL(k_tail_a, dim3(pl.nchunks, pl.Wb, T), dim3(pl.chunk), (size_t)pl.chunk * 128, s, y.L, y.counts, a_out, pl.NB,
                       pl.chunk, g.logchunk, pl.Wb, pl.ls0, pl.ls1);
    if (pl.nchunks > 32) {
        // 2^15 buckets per window (c = 16) are 128 chunks: 32 at a time (288 tree slots per block), then the 4 groups
        ge_rec *mid = b.chunk + g.chunk2_off;
        const int Wg = pl.Wb * g.ngroups, per1 = g.per0 + 5;
        hipLaunchKernelGGL(k_tail_b, dim3(Wg, T), dim3(320), (size_t)32 * g.per0 * 128, s, b.chunk, mid, 32, 5, g.per0, Wg);
        hipLaunchKernelGGL(k_tail_b, dim3(pl.Wb, T), dim3(320), (size_t)g.ngroups * per1 * 128, s, mid, b.out, g.ngroups,
                           ilog2(g.ngroups), per1, pl.Wb);
    } else if (pl.nchunks > 1)
        hipLaunchKernelGGL(k_tail_b, dim3(pl.Wb, T), dim3(320), (size_t)pl.nchunks * g.per0 * 128, s, b.chunk, b.out, pl.nchunks,
                           g.lognch, g.per0, pl.Wb);
}

int msm_submit(bpg_ctx *ctx, const msm_seg *segs, int nseg, const void *d_scalars, size_t n, int T, bool scalars_on_host,
               bpg_msm_job **out_job, msm_fixed_base fb, size_t sc_stride) {
    if (!ctx || !segs || !out_job || T < 1 || nseg < 1 || nseg > 4) return BPG_EINVAL;
    if (sc_stride && scalars_on_host) return BPG_EINVAL;
    if (n >= ((size_t)1 << 28) || (fb.c && (n >= ((size_t)1 << FB_WSHIFT) || fb.wstride >= ((size_t)1 << 32)))) return BPG_EINVAL;
    msm_maps mm{};
    int rc;
    if ((rc = msm_make_maps(segs, nseg, n, sc_stride, mm))) return rc;
    bpg_lane *lane = lane_of(ctx);
    if (!lane) return BPG_EDEVICE;
    bpg_msm_job *j = acquire_job(lane);
    if (!j) return BPG_EDEVICE;
    // The job is ours from here on.  Whatever fails below (a device allocation, a HIP call), it goes back to the
    // lane's pool: its stream orders anything already launched before the job's next use, and its buffers stay
    // owned by it.  *out_job is only set on success.
    j->busy.store(true, std::memory_order_release);
    struct Release {
        bpg_msm_job *j;
        ~Release() {
            if (j) j->busy.store(false, std::memory_order_release);
        }
    } release{j};
    msm_plan pl = make_plan(n, T, ctx->tune, fb.c, fb.density, ctx->resident_lanes);
    j->plan = pl;
    msm_geom g{};
    if ((rc = msm_geometry(pl, n, T, g))) return rc;
    if ((rc = grow((void **)&j->d_out, &j->cap_out, g.b_out))) return rc;
    if (g.b_out > j->cap_hout) {
        if (j->h_out) (void)hipHostFree(j->h_out);
        j->h_out = nullptr;
        j->cap_hout = 0;
        size_t want = g.b_out * 2;
        BPG_CHECK_HIP(hipHostMalloc((void **)&j->h_out, want, hipHostMallocDefault));
        j->cap_hout = want;
    }
    hipStream_t s = j->stream;
    BPG_CHECK_HIP(hipEventRecord(j->ev[0], s));
    if (n == 0) {
        BPG_CHECK_HIP(hipEventRecord(j->ev[1], s));
        BPG_CHECK_HIP(hipEventRecord(j->ev[2], s));
        BPG_CHECK_HIP(hipEventRecord(j->ev[3], s));
        release.j = nullptr;
        *out_job = j;
        return BPG_OK;
    }
    if ((rc = grow((void **)&j->d_digits, &j->cap_digits, g.b_digits))) return rc;
    if ((rc = grow((void **)&j->d_entries, &j->cap_entries, g.b_entries))) return rc;
    if ((rc = grow((void **)&j->d_counts, &j->cap_counts, g.b_counts))) return rc;
    if ((rc = grow((void **)&j->d_tmp, &j->cap_tmp, g.b_tmp))) return rc;
    if ((rc = grow((void **)&j->d_partials, &j->cap_partials, g.b_partials))) return rc;
    if ((rc = grow((void **)&j->d_chunk, &j->cap_chunk, g.b_chunk))) return rc;
    const uint32_t *d_sc = (const uint32_t *)d_scalars;
    if (scalars_on_host) {
        if ((rc = grow(&j->d_stage, &j->cap_stage, (size_t)T * n * 32))) return rc;
        BPG_CHECK_HIP(hipMemcpyAsync(j->d_stage, d_scalars, (size_t)T * n * 32, hipMemcpyHostToDevice, s));
        d_sc = (const uint32_t *)j->d_stage;
    }
    // the tail's few KB of output go straight to the job's pinned host buffer (device-visible at the same address): one
    // command less per MSM on the stream, and no hop through a copy engine between the last kernel and the host's wait
    static const bool zero_copy = getenv("BPG_NO_ZEROCOPY_OUT") == nullptr;
    const bool direct_out = zero_copy && g.b_out <= ((size_t)1 << 18);
    const msm_bufs b{j->d_digits, j->d_entries, j->d_counts, j->d_tmp, j->d_partials, j->d_chunk, direct_out ? j->h_out : j->d_out};
    const msm_layout y = msm_lay_out(g, b);
    msm_enqueue_sort(s, pl, g, b, y, d_sc, n, T, mm.smap, fb.c ? 1 : 0);
    // Accumulate passes of one lane's jobs run one after another: two of them at once only time-slice the
    // VALU, while the sort / merge / tail kernels of a later job do overlap usefully with an earlier job's pass.
    if (lane->acc_gate_job && lane->acc_gate_job != j) BPG_CHECK_HIP(hipStreamWaitEvent(s, lane->acc_gate_job->ev[2], 0));
    msm_enqueue_accumulate(s, pl, g, b, y, mm.map, nseg > 1, (uint32_t)fb.wstride, j->ev[1], j->ev[2]);
    lane->acc_gate_job = j;
    msm_enqueue_merge_tail(s, pl, g, b, y, T);
    BPG_CHECK_HIP(hipGetLastError());
    if (!direct_out) BPG_CHECK_HIP(hipMemcpyAsync(j->h_out, j->d_out, g.b_out, hipMemcpyDeviceToHost, s));
    BPG_CHECK_HIP(hipEventRecord(j->ev[3], s));
    release.j = nullptr;
    *out_job = j;
    return BPG_OK;
}

// ------------------------------------------------------------------ the entry point that allocates nothing
// For callers that cannot call hipMalloc (code under HIP graph capture, tight serving loops, harnesses that forbid
// it).  The same kernels as msm_submit(), with the workspace carved out of a file-scope device array instead of a
// job's growable buffers, everything enqueued on the caller's stream, no events, no lanes, no host buffer: the
// window combination that msm_wait() does on the host runs in k_tail_final on the device and the 32-byte encoding
// is written to the caller's device pointer.  Each GPU holds its own copy of the array for the life of the process.
constexpr size_t NOALLOC_MAX_POINTS = BPG_MSM_NOALLOC_MAX_POINTS;
constexpr size_t NOALLOC_SCRATCH_BYTES = (size_t)20 << 20;        // covers n = 16384 at any window the planner picks (checked per call)
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

// `scratch` in host code is a stand-in, not a device address: the per-device address comes from the runtime, once
// per context (= per device), with that device current (called by bpg_init).
int msm_noalloc_init(bpg_ctx *ctx) {
    void *ptr = nullptr;
    BPG_CHECK_HIP(hipSetDevice(ctx->device));
    BPG_CHECK_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    ctx->noalloc_scratch = ptr;
    ctx->noalloc_scratch_bytes = NOALLOC_SCRATCH_BYTES;
    return BPG_OK;
}

// One wave per target.  Lane w < Wb takes bucket set w = [total, S_0 .. S_{c-2}] to V_w = total + sum_j 2^j S_j
// (Horner over the bit sums), then lane 0 walks sum_w 2^(c w) V_w from the top window down and compresses.
// (One dependent chain of c * Wb doublings: about a millisecond on one lane.  msm_wait() does this on a host core.)
__global__ void __launch_bounds__(64) k_tail_final(const ge_rec *__restrict__ sets, int C, int Wb, uint8_t *__restrict__ out32) {
    __shared__ ge_rec v[64];
    const int w = threadIdx.x;
    const ge_rec *base = sets + (size_t)blockIdx.x * Wb * C;
    if (w < Wb) {
        ge acc, t;
        ge_identity(acc);
        for (int j = C - 2; j >= 0; j--) {
            ge_double(acc, acc);
            load_ge(t, base + (size_t)w * C + 1 + j);
            ge_add(acc, acc, t);
        }
        load_ge(t, base + (size_t)w * C);
        ge_add(acc, acc, t);
        store_ge(&v[w], acc);
    }
    __syncthreads();
    if (w != 0) return;
    // Everything below depends on wave-uniform values only (lane 0, LDS addresses without a lane term), and the
    // compiler then selects the whole chain of field arithmetic for the scalar unit and spends seven minutes in
    // "SI Fix SGPR copies" moving it back to the vector unit.  An offset it cannot see through (always zero, in a
    // VGPR) makes the loads, and with them the arithmetic, per-lane from the start.
    int lane_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
    ge acc, t;
    ge_identity(acc);
    for (int k = Wb - 1; k >= 0; k--) {
        if (k != Wb - 1)
            for (int d = 0; d < C; d++) ge_double(acc, acc);
        load_ge(t, &v[k + lane_zero]);
        ge_add(acc, acc, t);
    }
    uint32_t enc[8];
    ristretto_encode_w(enc, acc);
    uint4 *o = reinterpret_cast<uint4 *>(out32 + 32 * (size_t)blockIdx.x);
    o[0] = make_uint4(enc[0], enc[1], enc[2], enc[3]);
    o[1] = make_uint4(enc[4], enc[5], enc[6], enc[7]);
}

int msm_noalloc(bpg_ctx *ctx, const msm_seg *seg, const void *d_scalars, size_t n, void *d_out32, hipStream_t s) {
    if (!ctx || !seg || !d_out32 || (n && !d_scalars)) return BPG_EINVAL;
    if (n > NOALLOC_MAX_POINTS) {
        set_last_error_msg("bpg_msm_noalloc: more than BPG_MSM_NOALLOC_MAX_POINTS points (use bpg_msm_submit)");
        return BPG_EINVAL;
    }
    if (!ctx->noalloc_scratch) return BPG_EDEVICE;
    BPG_CHECK_HIP(hipSetDevice(ctx->device));
    if (n == 0) {                                          // the identity encodes as 32 zero bytes
        BPG_CHECK_HIP(hipMemsetAsync(d_out32, 0, 32, s));
        return BPG_OK;
    }
    msm_maps mm{};
    int rc;
    if ((rc = msm_make_maps(seg, 1, n, 0, mm))) return rc;
    // the automatic window for n, whatever BPG_MSM_* overrides the context carries: the bound on the workspace
    // (12.9 MiB at n = 16384) holds for the planner's own choice
    const msm_plan pl = make_plan(n, 1, msm_tuning(), 0, 1.f, ctx->resident_lanes);
    msm_geom g{};
    if ((rc = msm_geometry(pl, n, 1, g))) return rc;
    if (pl.Wb > 64) return BPG_EINVAL;                     // k_tail_final: one lane per window
    // carve the workspace (256-byte aligned pieces); everything the kernels read before writing is (re)initialised
    // by the kernels themselves (k_digits clears the sort's counters), so the last call's contents do not matter
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void *p = (uint8_t *)ctx->noalloc_scratch + off;
        off += (bytes + 255) & ~(size_t)255;
        return p;
    };
    msm_bufs b{};
    b.digits = (int16_t *)take(g.b_digits);
    b.entries = (uint32_t *)take(g.b_entries);
    b.counts = (uint32_t *)take(g.b_counts);
    b.tmp = take(g.b_tmp);
    b.partials = (ge_rec *)take(g.b_partials);
    b.chunk = (ge_rec *)take(g.b_chunk);
    b.out = (ge_rec *)take(g.b_out);
    if (off > ctx->noalloc_scratch_bytes) {
        set_last_error_msg("bpg_msm_noalloc: workspace for this size exceeds the static device array");
        return BPG_EINVAL;
    }
    const msm_layout y = msm_lay_out(g, b);
    msm_enqueue_sort(s, pl, g, b, y, (const uint32_t *)d_scalars, n, 1, mm.smap, 0);
    msm_enqueue_accumulate(s, pl, g, b, y, mm.map, false, 0u);
    msm_enqueue_merge_tail(s, pl, g, b, y, 1);
    hipLaunchKernelGGL(k_tail_final, dim3(1), dim3(64), 0, s, (const ge_rec *)b.out, pl.outs, pl.Wb, (uint8_t *)d_out32);
    BPG_CHECK_HIP(hipGetLastError());
    return BPG_OK;
}

// Folded generators (see k_fold_hist): rows scalar rows [s_G (np) | s_H (np)] `stride` scalars apart, the fixed-base
// tables of G[0..np) and H[0..np) in segs[0], segs[1]; writes rows * 2m affine Niels records
// [row][G'_0 .. G'_{m-1} | H'_0 .. H'_{m-1}] to d_tab (device).  Asynchronous: msm_fold_wait(job).
int msm_fold_submit(bpg_ctx *ctx, const msm_seg segs[2], const void *d_scalars, size_t stride, int rows, size_t np, size_t m,
                    msm_fixed_base fb, aniels_rec *d_tab, bpg_msm_job **out_job) {
    if (!ctx || !segs || !d_scalars || !d_tab || !out_job || rows < 1) return BPG_EINVAL;
    if (fb.c < 2 || fb.c > 14 || !fb.wstride) return BPG_EINVAL;                      // three 5-bit pieces cover |d| <= 2^13
    const fold_shape fs = fb.c <= 7 ? fold_shape{1, fb.c - 1} : fold_shape{3, 5};
    if (np < m || m < 32 || (m & (m - 1)) || (np & (np - 1)) || np > ((size_t)1 << (FB_WSHIFT - 1))) return BPG_EINVAL;
    if (segs[0].len != np || segs[1].len != np || !segs[0].base || !segs[1].base) return BPG_EINVAL;
    const size_t n = 2 * np;
    if (stride < n || stride >= ((size_t)1 << 32)) return BPG_EINVAL;
    bpg_lane *lane = lane_of(ctx);
    if (!lane) return BPG_EDEVICE;
    bpg_msm_job *j = acquire_job(lane);
    if (!j) return BPG_EDEVICE;
    const int c = fb.c, W = windows_for(c);
    const int S0 = 32, LS1 = 4;            // a 7-bit bucket holds 18 entries on average: one segment, nothing to merge
    const size_t ntargets = (size_t)rows * 2 * m, sets = ntargets * fs.P;
    const size_t NE = sets << fs.PB;                       // buckets
    const size_t M = (size_t)rows * fs.P * W * n;          // upper bound on entries
    if (W > (int)FB_WMASK + 1 || n > ((size_t)1 << FB_WSHIFT) || (((m * fs.P) << fs.PB) & 255)) return BPG_EINVAL;
    if (M >= ((size_t)1 << 32) || NE >= ((size_t)1 << 30)) return BPG_EINVAL;
    // merge passes: a bucket holds at most (np / m) * W entries
    const size_t worst = (np / m) * (size_t)W;
    int R = 0;
    while (((size_t)S0 << (LS1 * R)) < worst) R++;
    const int nlevels = R + 1;
    if (nlevels > MAX_LEVELS) return BPG_EINVAL;
    int tile = 4096;
    if ((size_t)tile > np) tile = (int)np;                 // a tile never straddles the G / H boundary
    const int ntiles = (int)(n / tile);
    const size_t nblk = (NE + 1023) / 1024;
    const size_t NBT = NE / 256;                           // bins of 256 buckets
    const uint32_t CH = 12288;
    const int LOWN = 256;
    const size_t NCH = M / CH + NBT;
    size_t cap_lv[MAX_LEVELS], part_off[MAX_LEVELS], part_total = 0;
    cap_lv[0] = M / (size_t)S0 + NE + 1;
    for (int r = 1; r < nlevels; r++) {
        const size_t by_count = ((M / (size_t)S0) >> (r * LS1)) + NE + 1;
        cap_lv[r] = cap_lv[r - 1] / 2 + 1 < by_count ? cap_lv[r - 1] / 2 + 1 : by_count;
    }
    for (int r = 0; r < nlevels; r++) {
        part_off[r] = part_total;
        part_total += cap_lv[r];
    }
    const size_t lvl_words = NE + nblk + 1;
    const size_t map_words = cap_lv[0] + (nlevels > 1 ? cap_lv[1] : 0);
    int rc;
    if ((rc = grow((void **)&j->d_out, &j->cap_out, sets * 6 * sizeof(ge_rec)))) return rc;
    if ((rc = grow((void **)&j->d_digits, &j->cap_digits, (size_t)rows * W * n * sizeof(int16_t)))) return rc;
    if ((rc = grow((void **)&j->d_entries, &j->cap_entries, M * sizeof(uint32_t)))) return rc;
    if ((rc = grow((void **)&j->d_counts, &j->cap_counts,
                   (2 * NE + (size_t)(nlevels + 1) * lvl_words + map_words + 4 * NBT + NCH * (size_t)LOWN + 16) * sizeof(uint32_t))))
        return rc;
    if ((rc = grow((void **)&j->d_tmp, &j->cap_tmp, M * sizeof(uint2) + 64))) return rc;
    if ((rc = grow((void **)&j->d_partials, &j->cap_partials, part_total * sizeof(ge_rec)))) return rc;
    j->busy = true;
    // a HIP failure below must not leave the job marked busy for ever
#define FOLD_CHECK_HIP(expr)                                  \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) {                               \
            bpg::set_last_error(#expr, _e);                   \
            j->busy.store(false, std::memory_order_release);  \
            return BPG_EDEVICE;                               \
        }                                                     \
    } while (0)
    j->plan = msm_plan{};
    j->plan.n = 0;                                         // nothing for msm_wait to fold on the host
    *out_job = j;
    hipStream_t s = j->stream;
    FOLD_CHECK_HIP(hipEventRecord(j->ev[0], s));
    seg_map map{};
    sc_map smap{};
    for (int k = 0; k < 4; k++) {
        map.base[k] = k < 2 ? segs[k].base - (size_t)k * np : nullptr;
        map.len[k] = k < 2 ? (uint32_t)((k + 1) * np) : 0xffffffffu;
        smap.off[k] = k < 2 ? (uint32_t)segs[k].sc_off : 0u;
        smap.end[k] = map.len[k];
    }
    smap.stride = stride;
    uint32_t *counts = j->d_counts, *ent_off = counts + NE;
    uint32_t *lvl_base = ent_off + NE;
    auto lv_local = [&](int idx) { return lvl_base + (size_t)idx * lvl_words; };
    auto lv_tot = [&](int idx) { return lvl_base + (size_t)idx * lvl_words + NE; };
    uint32_t *map0 = lvl_base + (size_t)(nlevels + 1) * lvl_words, *map1 = nlevels > 1 ? map0 + cap_lv[0] : nullptr;
    uint32_t *bin_counts = map0 + map_words, *bin_fill = bin_counts + NBT, *bin_off = bin_fill + NBT;
    uint32_t *chunk_off = bin_off + NBT + 1, *chunk_hist = chunk_off + NBT + 1;
    uint2 *tmp = (uint2 *)j->d_tmp;
    dispatch_digits(c, s, (const uint32_t *)d_scalars, j->d_digits, n, (size_t)rows * n, smap, bin_counts, (uint32_t)(2 * NBT));
    dim3 tgrid(ntiles, W, rows);
    const size_t bin_lds = (((m * fs.P) << fs.PB) >> 8) * sizeof(uint32_t);
    hipLaunchKernelGGL(k_fold_hist, tgrid, dim3(512), bin_lds, s, j->d_digits, bin_counts, n, W, (uint32_t)np, (uint32_t)m, tile, fs);
    hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, s, bin_counts, bin_off, chunk_off, (uint32_t)NBT, CH);
    hipLaunchKernelGGL(k_fold_scatter, tgrid, dim3(512), bin_lds, s, j->d_digits, bin_off, bin_fill, tmp, n, W, (uint32_t)np,
                       (uint32_t)m, tile, fs);
    hipLaunchKernelGGL(k_chunk_count<false>, dim3((unsigned)NCH), dim3(256), 0, s, bin_off, chunk_off, tmp, chunk_hist, counts, ent_off,
                       j->d_entries, (uint32_t)NBT, CH, LOWN);
    hipLaunchKernelGGL(k_chunk_offsets, dim3((unsigned)(NBT < 512 ? NBT : 512)), dim3(256), 0, s, bin_off, chunk_off, chunk_hist, counts,
                       ent_off, (uint32_t)NBT, LOWN);
    hipLaunchKernelGGL(k_chunk_place<false>, dim3((unsigned)(NCH < 2048 ? NCH : 2048)), dim3(256), 0, s, bin_off, chunk_off, tmp, chunk_hist,
                       ent_off, j->d_entries, (uint32_t)NBT, CH, LOWN);
    hipLaunchKernelGGL(k_scan_a, dim3((unsigned)nblk), dim3(256), 0, s, counts, lvl_base, lvl_words, NE, nlevels + 1, S0, LS1);
    hipLaunchKernelGGL(k_scan_b, dim3(nlevels + 1), dim3(1024), 0, s, lvl_base, lvl_words, NE, nblk);
    levels_dev L{};
    L.nlevels = nlevels;
    for (int r = 0; r < nlevels; r++) {
        L.lv[r] = level_ref{lv_local(r + 1), lv_tot(r + 1)};
        L.parts[r] = j->d_partials + part_off[r];
    }
    hipLaunchKernelGGL(k_segmap, dim3((unsigned)((NE + 255) / 256)), dim3(256), 0, s, counts, L.lv[0], L.lv[nlevels > 1 ? 1 : 0],
                       map0, map1, (uint32_t)NE, S0, LS1);
    if (lane->acc_gate_job && lane->acc_gate_job != j) FOLD_CHECK_HIP(hipStreamWaitEvent(s, lane->acc_gate_job->ev[2], 0));
    FOLD_CHECK_HIP(hipEventRecord(j->ev[1], s));
    {
        const unsigned blocks = (unsigned)((cap_lv[0] + 255) / 256), chunk = 0x7fffffff;
        for (unsigned b0 = 0; b0 < blocks; b0 += chunk)
            hipLaunchKernelGGL((k_accumulate<true>), dim3(blocks - b0 < chunk ? blocks - b0 : chunk), dim3(256), 0, s, j->d_entries, counts,
                               ent_off, L.lv[0], map0, j->d_partials + part_off[0], (uint32_t)nblk, S0, map, (uint32_t)fb.wstride, (uint32_t)b0);
    }
    FOLD_CHECK_HIP(hipEventRecord(j->ev[2], s));
    lane->acc_gate_job = j;
    for (int r = 1; r < nlevels; r++) {
        const uint32_t *mp = r == 1 ? (const uint32_t *)map1 : (const uint32_t *)nullptr;
        hipLaunchKernelGGL(k_reduce, dim3((unsigned)((cap_lv[r] + 255) / 256)), dim3(256), 0, s, counts, L.lv[r - 1], L.lv[r], mp,
                           j->d_partials + part_off[r - 1], j->d_partials + part_off[r], (uint32_t)nblk, (uint32_t)NE, r, S0, LS1);
    }
    hipLaunchKernelGGL(k_fold_sets, dim3((unsigned)((sets + 255) / 256)), dim3(256), 0, s, L, counts, (uint32_t)sets, j->d_out, S0, LS1, fs.PB);
    hipLaunchKernelGGL(k_fold_finish, dim3((unsigned)((ntargets + 255) / 256)), dim3(256), 0, s, j->d_out, (uint32_t)ntargets, d_tab, fs);
    FOLD_CHECK_HIP(hipGetLastError());
    FOLD_CHECK_HIP(hipEventRecord(j->ev[3], s));
    return BPG_OK;
#undef FOLD_CHECK_HIP
}

int msm_fold_wait(bpg_msm_job *j) {
    if (!j || !j->busy) return BPG_EINVAL;
    BPG_CHECK_HIP(wait_event(j->lane->ctx, j->ev[3]));
    j->busy.store(false, std::memory_order_release);
    return BPG_OK;
}

int msm_wait(bpg_msm_job *j, uint8_t *out, bool partial) {
    if (!j || !j->busy || !out) return BPG_EINVAL;
    BPG_CHECK_HIP(wait_event(j->lane->ctx, j->ev[3]));
    float acc_ms = 0.f, tot_ms = 0.f;
    (void)hipEventElapsedTime(&acc_ms, j->ev[1], j->ev[2]);
    (void)hipEventElapsedTime(&tot_ms, j->ev[0], j->ev[3]);
    j->lane->last_acc_us = acc_ms * 1000.f;
    j->lane->last_total_us = tot_ms * 1000.f;
    HostPhase hp_tail(HostPhase::TAIL_HOST);
    const msm_plan &pl = j->plan;
    for (int t = 0; t < pl.T; t++) {
        ge acc;
        ge_identity(acc);
        if (pl.n != 0) {
            const int C = pl.c, outs = pl.outs;
            const ge_rec *base = j->h_out + (size_t)t * pl.Wb * outs;
            // out layout per bucket set: [total, S_0, ..., S_{C-2}]; total and S_0 weigh 2^(C w), S_j weighs 2^(C w + j)
            // (fixed-base: one set, w = 0 -- the window weights are in the table)
            for (int pos = C * (pl.Wb - 1) + C - 2; pos >= 0; pos--) {
                ge_double(acc, acc);
                int w = pos / C, jbit = pos - w * C;
                if (jbit <= C - 2) {
                    ge s;
                    ge_load(s, base[(size_t)w * outs + 1 + jbit]);
                    ge_add(acc, acc, s);
                    if (jbit == 0) {
                        ge_load(s, base[(size_t)w * outs]);
                        ge_add(acc, acc, s);
                    }
                }
            }
        }
        if (partial) {
            // un-normalised extended coordinates X | Y | Z | T, 4 x 32 B little-endian
            fe_tobytes(out + 128 * t, acc.X);
            fe_tobytes(out + 128 * t + 32, acc.Y);
            fe_tobytes(out + 128 * t + 64, acc.Z);
            fe_tobytes(out + 128 * t + 96, acc.T);
        } else {
            ristretto_encode(out + 32 * t, acc);
        }
    }
    j->busy.store(false, std::memory_order_release);
    return BPG_OK;
}

// sum of k partial results (128-B extended points as written by msm_wait(partial)) -> compressed
int combine_partials(const uint8_t *ext128, size_t k, uint8_t out32[32]) {
    ge acc;
    ge_identity(acc);
    for (size_t i = 0; i < k; i++) {
        ge p;
        fe_frombytes(p.X, ext128 + 128 * i);
        fe_frombytes(p.Y, ext128 + 128 * i + 32);
        fe_frombytes(p.Z, ext128 + 128 * i + 64);
        fe_frombytes(p.T, ext128 + 128 * i + 96);
        // reject anything that is not a curve point with consistent T: -X^2 + Y^2 = Z^2 + d T^2, XY = ZT
        fe xx, yy, zz, tt, l, r, xy, zt;
        fe_sq(xx, p.X); fe_sq(yy, p.Y); fe_sq(zz, p.Z); fe_sq(tt, p.T);
        fe_sub(l, yy, xx);
        fe_mul(r, tt, fe_d());
        fe_add(r, r, zz);
        fe_mul(xy, p.X, p.Y);
        fe_mul(zt, p.Z, p.T);
        if (!fe_eq(l, r) || !fe_eq(xy, zt) || fe_iszero(p.Z)) return BPG_EFORMAT;
        ge_add(acc, acc, p);
    }
    ristretto_encode(out32, acc);
    return BPG_OK;
}

}  // namespace bpg
