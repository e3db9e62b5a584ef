// tools/irate.hip -- issue cost of single gfx950 VALU instructions, in SIMD cycles per wave-instruction.
// Each kernel runs one instruction in eight independent register chains per lane, unrolled, 8 waves per SIMD, for
// several milliseconds; cycles = (CUs x 4 SIMDs x clock) / (wave-instructions per second).  The field-arithmetic
// design is priced with these numbers: a mixed addition's time is the sum of its
// instructions' costs, and which instructions are half rate decides the limb representation.
// Build: hipcc -O3 --offload-arch=gfx950 tools/irate.hip -o tools/irate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                   \
    do {                                                           \
        hipError_t e = (x);                                        \
        if (e != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); \
            exit(1);                                               \
        }                                                          \
    } while (0)

constexpr int ITERS = 16384;

// BODY uses registers r0..r7 (32-bit, "+v"), p0..p3 (64-bit, "+v"), x, y (32-bit inputs)
#define KERNEL(NAME, NINSTR, ASM)                                                                                      \
    __global__ void __launch_bounds__(256) NAME(uint32_t *out, uint32_t a, uint32_t b) {                              \
        uint32_t r0 = a + threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6,  \
                 r7 = r0 + 7, x = b | 1u, y = a ^ 0x9e3779b9u;                                                        \
        uint64_t p0 = r0, p1 = r1, p2 = r2, p3 = r3, p4 = r4, p5 = r5, p6 = r6, p7 = r7;                              \
        for (int i = 0; i < ITERS; i++) {                                                                              \
            asm volatile(ASM ASM ASM ASM                                                                               \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(p0),  \
                           "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)                       \
                         : "v"(x), "v"(y)                                                                              \
                         : "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",  \
                           "s52", "s53", "s54", "s55");                                                                \
        }                                                                                                              \
        out[blockIdx.x * 256 + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^                                 \
                                              (uint32_t)(p0 ^ p1 ^ p2 ^ p3 ^ p4 ^ p5 ^ p6 ^ p7);                      \
    }                                                                                                                  \
    static const int NAME##_n = 4 * (NINSTR);

// eight copies of a one-register instruction
#define R8(fmt_a, fmt_b)                                                                                     \
    fmt_a "%0" fmt_b "%0\n" fmt_a "%1" fmt_b "%1\n" fmt_a "%2" fmt_b "%2\n" fmt_a "%3" fmt_b "%3\n" fmt_a "%4" \
          fmt_b "%4\n" fmt_a "%5" fmt_b "%5\n" fmt_a "%6" fmt_b "%6\n" fmt_a "%7" fmt_b "%7\n"
#define R8S(fmt_a, fmt_b, sfx)                                                                                      \
    fmt_a "%0" fmt_b "%0" sfx "\n" fmt_a "%1" fmt_b "%1" sfx "\n" fmt_a "%2" fmt_b "%2" sfx "\n" fmt_a "%3" fmt_b "%3" sfx "\n" \
    fmt_a "%4" fmt_b "%4" sfx "\n" fmt_a "%5" fmt_b "%5" sfx "\n" fmt_a "%6" fmt_b "%6" sfx "\n" fmt_a "%7" fmt_b "%7" sfx "\n"
#define P8(fmt_a, fmt_b)                                                                                        \
    fmt_a "%8" fmt_b "%8\n" fmt_a "%9" fmt_b "%9\n" fmt_a "%10" fmt_b "%10\n" fmt_a "%11" fmt_b "%11\n" fmt_a    \
          "%12" fmt_b "%12\n" fmt_a "%13" fmt_b "%13\n" fmt_a "%14" fmt_b "%14\n" fmt_a "%15" fmt_b "%15\n"

KERNEL(k_add_u32, 8, R8("v_add_u32 ", ", %16, "))
KERNEL(k_mov_b32, 8, R8("v_mov_b32 ", ", "))
KERNEL(k_and_b32, 8, R8("v_and_b32 ", ", %16, "))
KERNEL(k_add3_u32, 8, R8("v_add3_u32 ", ", %16, %17, "))
KERNEL(k_alignbit, 8, R8("v_alignbit_b32 ", ", %16, %17, "))
KERNEL(k_bfe_u32, 8, R8("v_bfe_u32 ", ", %16, 3, "))
KERNEL(k_lshl_add_u32, 8, R8("v_lshl_add_u32 ", ", %16, 3, "))
KERNEL(k_and_or_b32, 8, R8("v_and_or_b32 ", ", %16, %17, "))
KERNEL(k_mul_u32_u24, 8, R8("v_mul_u32_u24 ", ", %16, "))
KERNEL(k_mad_u32_u24, 8, R8("v_mad_u32_u24 ", ", %16, %17, "))
KERNEL(k_mul_lo_u32, 8, R8("v_mul_lo_u32 ", ", %16, "))
KERNEL(k_mul_hi_u32, 8, R8("v_mul_hi_u32 ", ", %16, "))
// carries: out only (vcc), out only (sgpr pair), in + out chains (vcc; two chains on two sgpr pairs), select by an sgpr mask
KERNEL(k_add_co_vcc, 8, R8("v_add_co_u32 ", ", vcc, %16, "))
KERNEL(k_add_co_sgpr, 8, R8("v_add_co_u32 ", ", s[40:41], %16, "))
KERNEL(k_addc_co_vcc, 8, R8S("v_addc_co_u32 ", ", vcc, %16, ", ", vcc") /* reads the carry of the one before: a true chain */)
KERNEL(k_addc_co_2chains, 8,
       "v_addc_co_u32 %0, s[40:41], %16, %0, s[40:41]\nv_addc_co_u32 %1, s[42:43], %16, %1, s[42:43]\n"
       "v_addc_co_u32 %2, s[44:45], %16, %2, s[44:45]\nv_addc_co_u32 %3, s[40:41], %16, %3, s[40:41]\n"
       "v_addc_co_u32 %4, s[42:43], %16, %4, s[42:43]\nv_addc_co_u32 %5, s[44:45], %16, %5, s[44:45]\n"
       "v_addc_co_u32 %6, s[40:41], %16, %6, s[40:41]\nv_addc_co_u32 %7, s[42:43], %16, %7, s[42:43]\n")
KERNEL(k_cndmask_sgpr, 8, R8S("v_cndmask_b32 ", ", %16, ", ", vcc") /* vcc as mask */)
KERNEL(k_cmp_vcc, 8, R8("v_cmp_lt_u32 vcc, ", ", "))
// 64-bit
KERNEL(k_mad_u64_u32, 8, P8("v_mad_u64_u32 ", ", s[40:41], %16, %17, "))
KERNEL(k_mad_u64_u32_vccout, 8, P8("v_mad_u64_u32 ", ", vcc, %16, %17, "))
KERNEL(k_mad_i64_i32, 8, P8("v_mad_i64_i32 ", ", s[40:41], %16, %17, "))
KERNEL(k_lshl_add_u64, 8, P8("v_lshl_add_u64 ", ", %8, 0, ") /* p = (p0 << 0) + p */)
KERNEL(k_lshrrev_b64, 8, P8("v_lshrrev_b64 ", ", 3, "))
KERNEL(k_lshlrev_b64, 8, P8("v_lshlrev_b64 ", ", 3, "))
KERNEL(k_ashrrev_i64, 8, P8("v_ashrrev_i64 ", ", 3, "))
KERNEL(k_mov_b64, 8, P8("v_mov_b64 ", ", "))
KERNEL(k_fma_f64, 8, P8("v_fma_f64 ", ", %8, %8, "))
KERNEL(k_pk_add_u16, 8, R8("v_pk_add_u16 ", ", %16, "))
KERNEL(k_dot4_u32_u8, 8, R8("v_dot4_u32_u8 ", ", %16, %17, "))
// mixes
KERNEL(k_mix_mad_addc, 16,
       "v_mad_u64_u32 %8, s[40:41], %16, %17, %8\nv_addc_co_u32 %0, vcc, %16, %0, vcc\n"
       "v_mad_u64_u32 %9, s[40:41], %16, %17, %9\nv_addc_co_u32 %1, s[42:43], %16, %1, s[42:43]\n"
       "v_mad_u64_u32 %10, s[40:41], %16, %17, %10\nv_addc_co_u32 %2, s[44:45], %16, %2, s[44:45]\n"
       "v_mad_u64_u32 %11, s[40:41], %16, %17, %11\nv_addc_co_u32 %3, vcc, %16, %3, vcc\n"
       "v_mad_u64_u32 %12, s[40:41], %16, %17, %12\nv_addc_co_u32 %4, s[42:43], %16, %4, s[42:43]\n"
       "v_mad_u64_u32 %13, s[40:41], %16, %17, %13\nv_addc_co_u32 %5, s[44:45], %16, %5, s[44:45]\n"
       "v_mad_u64_u32 %14, s[40:41], %16, %17, %14\nv_addc_co_u32 %6, vcc, %16, %6, vcc\n"
       "v_mad_u64_u32 %15, s[40:41], %16, %17, %15\nv_addc_co_u32 %7, s[42:43], %16, %7, s[42:43]\n")
KERNEL(k_mix_mad_add, 16,
       "v_mad_u64_u32 %8, s[40:41], %16, %17, %8\nv_add_u32 %0, %16, %0\n"
       "v_mad_u64_u32 %9, s[40:41], %16, %17, %9\nv_add_u32 %1, %16, %1\n"
       "v_mad_u64_u32 %10, s[40:41], %16, %17, %10\nv_add_u32 %2, %16, %2\n"
       "v_mad_u64_u32 %11, s[40:41], %16, %17, %11\nv_add_u32 %3, %16, %3\n"
       "v_mad_u64_u32 %12, s[40:41], %16, %17, %12\nv_add_u32 %4, %16, %4\n"
       "v_mad_u64_u32 %13, s[40:41], %16, %17, %13\nv_add_u32 %5, %16, %5\n"
       "v_mad_u64_u32 %14, s[40:41], %16, %17, %14\nv_add_u32 %6, %16, %6\n"
       "v_mad_u64_u32 %15, s[40:41], %16, %17, %15\nv_add_u32 %7, %16, %7\n")

template <typename K>
static double run(K kernel, int blocks, uint32_t *buf) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, buf, 3u, 5u);
    CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < 3; r++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, buf, 3u, 5u);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv) {
    int bpc = argc > 1 ? atoi(argv[1]) : 8;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, blocks = cus * bpc;
    uint32_t *buf;
    CHECK(hipMalloc((void **)&buf, (size_t)blocks * 256 * 4));
    const double simd_cycles_per_ms = (double)cus * 4 * 2.4e6;      // at the 2.4 GHz nominal clock
    printf("{\"device\": \"%s\", \"cus\": %d, \"blocks_per_cu\": %d, \"nominal_clock_ghz\": 2.4, \"cycles_per_wave_instruction\": {\n", prop.gcnArchName, cus, bpc);
#define RUN(NAME)                                                                                             \
    {                                                                                                         \
        double ms = run(NAME, blocks, buf);                                                                   \
        double winstr = (double)blocks * 4 * ITERS * NAME##_n;                                                \
        printf("  \"%s\": %.2f,\n", #NAME + 2, simd_cycles_per_ms * ms / winstr);                             \
    }
    RUN(k_add_u32) RUN(k_mov_b32) RUN(k_and_b32) RUN(k_add3_u32) RUN(k_alignbit) RUN(k_bfe_u32) RUN(k_lshl_add_u32)
    RUN(k_and_or_b32) RUN(k_mul_u32_u24) RUN(k_mad_u32_u24) RUN(k_mul_lo_u32) RUN(k_mul_hi_u32) RUN(k_add_co_vcc)
    RUN(k_add_co_sgpr) RUN(k_addc_co_vcc) RUN(k_addc_co_2chains) RUN(k_cndmask_sgpr) RUN(k_cmp_vcc) RUN(k_mad_u64_u32)
    RUN(k_mad_u64_u32_vccout) RUN(k_mad_i64_i32) RUN(k_lshl_add_u64) RUN(k_lshrrev_b64) RUN(k_lshlrev_b64) RUN(k_ashrrev_i64)
    RUN(k_mov_b64) RUN(k_fma_f64) RUN(k_pk_add_u16) RUN(k_dot4_u32_u8) RUN(k_mix_mad_addc) RUN(k_mix_mad_add)
    printf("  \"_\": 0}}\n");
    return 0;
}
