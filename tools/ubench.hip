// tools/ubench.hip -- integer-VALU microbenchmarks for gfx950.
// Measures the sustained rate of the instructions the field arithmetic is
// built from (v_mad_u64_u32, v_mul_lo/hi_u32, v_add_co/v_addc_co, v_fma_f64) and
// of whole field operations (fe_mul, fe_sq, mixed addition), so that the MSM
// roofline has a measured denominator.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#include "../bulletproofs_gadgets_amd/csrc/hot25519.h"

using namespace bpg;

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                 \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

// Every timed kernel runs for several milliseconds (launch ramp and tail are then < 1 %): round 1 timed 75-us
// kernels and read every rate about 25 % low.  ITERS scales the eight-chain primitives; the field /
// group loops take their own counts below.
constexpr int ITERS = 65536;
constexpr int FE_ITERS = 8192;      // fe_mul / fe_sq calls per lane
constexpr int GE_ITERS = 1024;      // mixed additions per lane

// 8 independent mad chains per lane, each with its own multiplicands (one v_mad_u64_u32 per
// chain step: with shared multiplicands the compiler computes the product once and adds it
// eight times, which is what an earlier version of this test measured by mistake)
__global__ void __launch_bounds__(256) k_mad64(uint64_t *out, uint32_t a, uint32_t b) {
    uint64_t acc[8];
    uint32_t x[8], y[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        acc[k] = k + threadIdx.x;
        x[k] = a * (2 * k + 3) + threadIdx.x;
        y[k] = b * (2 * k + 5) + blockIdx.x;
    }
#pragma unroll 8
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = (uint64_t)x[k] * y[k] + acc[k];
#pragma unroll
        for (int k = 0; k < 8; k++) asm volatile("" : "+v"(x[k]));      // opaque: no hoisting of x*y
    }
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s ^= acc[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_mullo(uint32_t *out, uint32_t a) {
    uint32_t acc[8];
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = a + k + threadIdx.x;
#pragma unroll 8
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = acc[k] * acc[k] + 1u;   // v_mul_lo_u32 + add (or v_mad_u32?)
    }
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s ^= acc[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_mulhi(uint32_t *out, uint32_t a) {
    uint32_t acc[8];
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = a + k + threadIdx.x;
#pragma unroll 8
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = __umulhi(acc[k], 0x9e3779b9u + k) ^ a;
    }
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s ^= acc[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_addc(uint32_t *out, uint32_t a) {
    uint32_t acc[8];
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = a + k + threadIdx.x;
#pragma unroll 8
    for (int i = 0; i < ITERS; i++) {
        uint32_t c = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = addc32(acc[k], a, c, c);
    }
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s ^= acc[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_add32(uint32_t *out, uint32_t a) {
    uint32_t acc[8];
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = a + k + threadIdx.x;
#pragma unroll 8
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = (acc[k] + a) ^ (acc[(k + 1) & 7]);   // 2 simple ops
    }
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s ^= acc[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_dfma(double *out, double a, double b) {
    double acc[8];
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = a + k + threadIdx.x;
#pragma unroll 8
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = __builtin_fma(acc[k], b, a);
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s += acc[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_femul(fe *out, const fe *in) {
    int i = blockIdx.x * 256 + threadIdx.x;
    fe x = in[i], y = in[i + 1];
    for (int k = 0; k < FE_ITERS / 2; k++) {
        fe_mul(x, x, y);
        fe_mul(y, y, x);
    }
    out[i] = x;
    out[i + 1] = y;
}

__global__ void __launch_bounds__(256) k_fesq(fe *out, const fe *in) {
    int i = blockIdx.x * 256 + threadIdx.x;
    fe x = in[i], y = in[i + 1];
    for (int k = 0; k < FE_ITERS / 2; k++) {
        fe_sq(x, x);
        fe_sq(y, y);
    }
    out[i] = x;
    out[i + 1] = y;
}

__global__ void __launch_bounds__(256) k_madd(fe *out, const fe *in) {
    int i = blockIdx.x * 256 + threadIdx.x;
    ge acc;
    ge_identity(acc);
    ge_aniels q;
    q.hYpX = in[i];
    q.hYmX = in[i + 1];
    q.Tdxy = in[i + 2];
    for (int k = 0; k < GE_ITERS; k++) ge_madd(acc, acc, q, (k & 1) != 0);
    out[i] = acc.X;
    out[i + 1] = acc.Y;
    out[i + 2] = acc.Z;
    out[i + 3] = acc.T;
}

// the hot-path forms (hot25519.h): zero-partner register pairs, rare-path ripples, sign by operand order
__global__ void __launch_bounds__(256) k_femulz(fe *out, const fe *in) {
    int i = blockIdx.x * 256 + threadIdx.x;
    fe x = in[i], y = in[i + 1];
    fe_zeros Z;
    fe_zeros_init(Z);
    for (int k = 0; k < FE_ITERS / 2; k++) {
        fez_mul(x, x, y, Z);
        fez_mul(y, y, x, Z);
    }
    out[i] = x;
    out[i + 1] = y;
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) k_maddz(fe *out, const fe *in) {
    int i = blockIdx.x * 256 + threadIdx.x;
    ge acc;
    ge_identity(acc);
    ge_aniels_signed q, qn;
    q.A = in[i];
    q.B = in[i + 1];
    q.C = in[i + 2];
    qn.A = q.B;
    qn.B = q.A;
    qn.C = in[i + 3];
    fe_zeros Z;
    fe_zeros_init(Z);
    for (int k = 0; k < GE_ITERS / 2; k++) {
        gez_madd(acc, acc, q, Z);
        gez_madd(acc, acc, qn, Z);
    }
    out[i] = acc.X;
    out[i + 1] = acc.Y;
    out[i + 2] = acc.Z;
    out[i + 3] = acc.T;
}
// the generated instruction stream (madd_gfx950.inc)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) k_maddasm_stamped(fe *out, const fe *in, uint64_t *stamps, int iters) {
    int i = blockIdx.x * 256 + threadIdx.x;
    ge acc;
    ge_identity(acc);
    ge_aniels_signed q;
    const fe *src = in + i;
    uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int k = 0; k < iters; k++) {
        q.A = src[0];                      // the table entry is loaded per addition, as in the real kernel (L1 / L2 hits here)
        q.B = src[1];
        q.C = src[2];
        gez_madd_asm(acc, q);
        src = in + i + (acc.X.v[0] & 1);   // keep the address opaque
    }
    uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[i] = acc.X;
    out[i + 1] = acc.Y;
    out[i + 2] = acc.Z;
    out[i + 3] = acc.T;
    if ((threadIdx.x & 63) == 0) {
        size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * w] = c1 - c0;
        stamps[2 * w + 1] = r1 - r0;
    }
}
// correctness on the device: each form of the mixed addition on the same random inputs, 24 additions deep; canonical
// words of the result go to memory and the host compares the three arrays
template <int FORM>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) k_madd_form(const fe *in, uint32_t *out, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    ge a;
    a.X = in[i];
    a.Y = in[i + 1];
    a.Z = in[i + 2];
    a.T = in[i + 3];
    fe_zeros Z;
    if (FORM == 1) fe_zeros_init(Z);
    for (int k = 0; k < 24; k++) {
        const bool neg = (k % 3) == 1;
        if (FORM == 0) {
            ge_aniels qq;
            qq.hYpX = in[i + 4 + k];
            qq.hYmX = in[i + 5 + k];
            qq.Tdxy = in[i + 6 + k];
            ge_madd(a, a, qq, neg);
        } else {
            ge_aniels_signed s;
            s.A = in[i + (neg ? 5 : 4) + k];
            s.B = in[i + (neg ? 4 : 5) + k];
            s.C = in[i + 6 + k];
            if (neg) fe_neg(s.C, s.C);
            if (FORM == 1) gez_madd(a, a, s, Z);
            else gez_madd_asm(a, s);
        }
    }
    const fe *f[4] = {&a.X, &a.Y, &a.Z, &a.T};
    for (int c = 0; c < 4; c++) {
        uint32_t w[8];
        fe_towords(w, *f[c]);
        for (int k = 0; k < 8; k++) out[(size_t)i * 32 + c * 8 + k] = w[k];
    }
}
// stamped like k_madd_stamped
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) k_maddz_stamped(fe *out, const fe *in, uint64_t *stamps, int iters) {
    int i = blockIdx.x * 256 + threadIdx.x;
    ge acc;
    ge_identity(acc);
    ge_aniels_signed q;
    q.A = in[i];
    q.B = in[i + 1];
    q.C = in[i + 2];
    fe_zeros Z;
    fe_zeros_init(Z);
    uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int k = 0; k < iters; k++) {
        gez_madd(acc, acc, q, Z);
        q.C.v[0] ^= acc.X.v[0];
    }
    uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[i] = acc.X;
    out[i + 1] = acc.Y;
    out[i + 2] = acc.Z;
    out[i + 3] = acc.T;
    if ((threadIdx.x & 63) == 0) {
        size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * w] = c1 - c0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

// ge_madd loop with in-kernel stamps: cycles (s_memtime) and 100 MHz ticks (s_memrealtime) per wave,
// written to a buffer nothing else reads.  Gives cycles per mixed addition at the launch's occupancy
// and the clock the chip holds under this load (MI355X_MICROARCH.md, DVFS give-back item 6).
__global__ void __launch_bounds__(256) k_madd_stamped(fe *out, const fe *in, uint64_t *stamps, int iters) {
    int i = blockIdx.x * 256 + threadIdx.x;
    ge acc;
    ge_identity(acc);
    ge_aniels q;
    q.hYpX = in[i];
    q.hYmX = in[i + 1];
    q.Tdxy = in[i + 2];
    uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int k = 0; k < iters; k++) {
        ge_madd(acc, acc, q, (k & 1) != 0);
        q.Tdxy.v[0] ^= acc.X.v[0];          // keep the operand varying
    }
    uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[i] = acc.X;
    out[i + 1] = acc.Y;
    out[i + 2] = acc.Z;
    out[i + 3] = acc.T;
    if ((threadIdx.x & 63) == 0) {
        size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * w] = c1 - c0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

template <typename F>
static double time_ms(F launch, int reps = 5) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    launch();
    CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; r++) {
        CHECK(hipEventRecord(a));
        launch();
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv) {
    int blocks_per_cu = argc > 1 ? atoi(argv[1]) : 8;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    int blocks = cus * blocks_per_cu;
    size_t nthreads = (size_t)blocks * 256;
    void *buf, *in;
    CHECK(hipMalloc(&buf, nthreads * 64 + 4096));
    CHECK(hipMalloc(&in, nthreads * 32 + 4096));
    std::vector<uint32_t> h((nthreads * 32 + 4096) / 4);
    uint32_t s = 12345;
    for (auto &x : h) {
        s = s * 1664525u + 1013904223u;
        x = s;
    }
    CHECK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"blocks_per_cu\": %d,\n", prop.gcnArchName, cus,
           prop.clockRate / 1000, blocks_per_cu);
    double ms;
    const double lanes = (double)nthreads;
    // anchors from the guide (MI355X_MICROARCH.md): 256 CUs x 4 SIMDs x 64 lanes x 2.4 GHz = 157.3 T lane-cycles/s;
    // a full-rate VALU instruction takes 2 cycles per wave (78.6 T lane-ops/s), v_fma_f64 4 cycles (39.3 T/s)
    const double lane_cycles_T = (double)cus * 4 * 64 * 2.4e9 / 1e12;
    printf(" \"anchor_simple_int32_Tops\": %.1f, \"anchor_fma_f64_Tops\": %.1f,\n", lane_cycles_T / 2, lane_cycles_T / 4);
    ms = time_ms([&] { hipLaunchKernelGGL(k_mad64, dim3(blocks), dim3(256), 0, 0, (uint64_t *)buf, 3u, 5u); });
    printf(" \"v_mad_u64_u32_Gops\": %.1f, \"v_mad_u64_u32_kernel_ms\": %.2f, \"v_mad_u64_u32_cycles_per_wave_instr\": %.2f,\n",
           lanes * ITERS * 8 / ms / 1e6, ms, lane_cycles_T * 1e12 / (lanes * ITERS * 8 / ms * 1e3));
    ms = time_ms([&] { hipLaunchKernelGGL(k_mullo, dim3(blocks), dim3(256), 0, 0, (uint32_t *)buf, 3u); });
    printf(" \"mul_lo_plus_add_Gops\": %.1f,\n", lanes * ITERS * 8 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_mulhi, dim3(blocks), dim3(256), 0, 0, (uint32_t *)buf, 3u); });
    printf(" \"mul_hi_plus_xor_Gops\": %.1f,\n", lanes * ITERS * 8 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_addc, dim3(blocks), dim3(256), 0, 0, (uint32_t *)buf, 3u); });
    printf(" \"v_addc_chain_Gops\": %.1f,\n", lanes * ITERS * 8 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_add32, dim3(blocks), dim3(256), 0, 0, (uint32_t *)buf, 3u); });
    printf(" \"simple_int32_Gops\": %.1f, \"simple_int32_kernel_ms\": %.2f,\n", lanes * ITERS * 16 / ms / 1e6, ms);
    ms = time_ms([&] { hipLaunchKernelGGL(k_dfma, dim3(blocks), dim3(256), 0, 0, (double *)buf, 1.0000001, 0.9999999); });
    printf(" \"v_fma_f64_Gops\": %.1f, \"v_fma_f64_kernel_ms\": %.2f,\n", lanes * ITERS * 8 / ms / 1e6, ms);
    ms = time_ms([&] { hipLaunchKernelGGL(k_femul, dim3(blocks), dim3(256), 0, 0, (fe *)buf, (const fe *)in); });
    printf(" \"fe_mul_G\": %.2f,\n", lanes * FE_ITERS / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_femulz, dim3(blocks), dim3(256), 0, 0, (fe *)buf, (const fe *)in); });
    printf(" \"fez_mul_G\": %.2f,\n", lanes * FE_ITERS / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_fesq, dim3(blocks), dim3(256), 0, 0, (fe *)buf, (const fe *)in); });
    printf(" \"fe_sq_G\": %.2f,\n", lanes * FE_ITERS / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_madd, dim3(blocks), dim3(256), 0, 0, (fe *)buf, (const fe *)in); });
    printf(" \"ge_madd_G\": %.3f, \"ge_madd_kernel_ms\": %.2f,\n", lanes * GE_ITERS / ms / 1e6, ms);
    ms = time_ms([&] { hipLaunchKernelGGL(k_maddz, dim3(blocks), dim3(256), 0, 0, (fe *)buf, (const fe *)in); });
    printf(" \"gez_madd_G\": %.3f, \"gez_madd_kernel_ms\": %.2f,\n", lanes * GE_ITERS / ms / 1e6, ms);
    // stamped mixed-addition loops at 1, 2, 3, 4 blocks per CU (blocks of 4 waves: one wave per SIMD each; how many
    // are resident at once depends on the kernel's VGPR count, printed by the build) and with the grid the primitives use
    {
        const int iters = 2000;
        uint64_t *stamps;
        CHECK(hipMalloc((void **)&stamps, (size_t)blocks * 4 * 16));
        std::vector<uint64_t> hs((size_t)blocks * 8);
        {
            const int ncmp = 1 << 18;
            uint32_t *d_res;
            CHECK(hipMalloc((void **)&d_res, (size_t)3 * ncmp * 128));
            hipLaunchKernelGGL(k_madd_form<0>, dim3(ncmp / 256), dim3(256), 0, 0, (const fe *)in, d_res, ncmp);
            hipLaunchKernelGGL(k_madd_form<1>, dim3(ncmp / 256), dim3(256), 0, 0, (const fe *)in, d_res + (size_t)ncmp * 32, ncmp);
            hipLaunchKernelGGL(k_madd_form<2>, dim3(ncmp / 256), dim3(256), 0, 0, (const fe *)in, d_res + (size_t)2 * ncmp * 32, ncmp);
            std::vector<uint32_t> hr((size_t)3 * ncmp * 32);
            CHECK(hipMemcpy(hr.data(), d_res, hr.size() * 4, hipMemcpyDeviceToHost));
            size_t bad1 = 0, bad2 = 0;
            for (size_t k = 0; k < (size_t)ncmp * 32; k++) {
                bad1 += hr[k] != hr[(size_t)ncmp * 32 + k];
                bad2 += hr[k] != hr[(size_t)2 * ncmp * 32 + k];
            }
            printf(" \"madd_forms_compared\": %d, \"gez_madd_mismatched_words\": %zu, \"gez_madd_asm_mismatched_words\": %zu,\n", ncmp, bad1, bad2);
            CHECK(hipFree(d_res));
        }
        for (int variant = 0; variant < 3; variant++) {
            printf(" \"%s\": [", variant == 2 ? "gez_madd_asm_stamped" : variant ? "gez_madd_stamped" : "ge_madd_stamped");
            const int bpcs[5] = {1, 2, 3, 4, blocks_per_cu};
            for (int v = 0; v < 5; v++) {
                int nb = cus * bpcs[v];
                auto launch = [&] {
                    if (variant == 2)
                        hipLaunchKernelGGL(k_maddasm_stamped, dim3(nb), dim3(256), 0, 0, (fe *)buf, (const fe *)in, stamps, iters);
                    else if (variant)
                        hipLaunchKernelGGL(k_maddz_stamped, dim3(nb), dim3(256), 0, 0, (fe *)buf, (const fe *)in, stamps, iters);
                    else
                        hipLaunchKernelGGL(k_madd_stamped, dim3(nb), dim3(256), 0, 0, (fe *)buf, (const fe *)in, stamps, iters);
                };
                for (int rep = 0; rep < 2; rep++) launch();
                hipEvent_t a, b;
                CHECK(hipEventCreate(&a));
                CHECK(hipEventCreate(&b));
                CHECK(hipEventRecord(a));
                launch();
                CHECK(hipEventRecord(b));
                CHECK(hipEventSynchronize(b));
                float ems;
                CHECK(hipEventElapsedTime(&ems, a, b));
                CHECK(hipMemcpy(hs.data(), stamps, (size_t)nb * 4 * 16, hipMemcpyDeviceToHost));
                std::vector<double> cyc, clk;
                for (int w = 0; w < nb * 4; w++) {
                    cyc.push_back((double)hs[2 * w] / iters);
                    clk.push_back((double)hs[2 * w] / (double)hs[2 * w + 1] * 100.0);
                }
                std::sort(cyc.begin(), cyc.end());
                std::sort(clk.begin(), clk.end());
                printf("%s{\"blocks_per_cu\": %d, \"cycles_per_madd_per_wave\": %.0f, \"clock_mhz\": %.0f, \"madd_G\": %.2f, \"kernel_ms\": %.1f}",
                       v ? ", " : "", bpcs[v], cyc[cyc.size() / 2], clk[clk.size() / 2], (double)nb * 256 * iters / ems / 1e6, ems);
            }
            printf("]%s\n", variant == 2 ? "" : ",");
        }
        printf("}\n");
    }
    return 0;
}
