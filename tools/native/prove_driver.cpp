// A native load generator for the proving path: T host threads, each proving R groups of K statements (the Edwards
// scalar_mul gadget over `nbits` committed bits) through the C ABI alone -- what a Rust or C++ prover service does.
// It answers one question bench.py cannot: how much of the per-process limit on proofs/s is the Python driver.
//
//   python tools/dump_statement.py 2427 gpurun_out/stmt.bin && tools/native/prove_driver gpurun_out/stmt.bin 64 3 8
//
// Statement file: base point (X|Y|Z|T, 4 x 32 bytes), expected point (128 bytes), u32 nbits, nbits x 32-byte values.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
#include <sys/resource.h>

#include "bpg.h"

static std::vector<uint8_t> g_base(128), g_expected(128), g_vals;
static uint32_t g_nbits;
static bpg_gens *g_gens;
static std::atomic<int> g_failed{0};

struct Xs {
    uint64_t s;
    uint64_t next() {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        return s;
    }
};
static void rng_cb(void *user, uint8_t *out, size_t n) {
    Xs *x = (Xs *)user;
    for (size_t i = 0; i < n; i++) out[i] = (uint8_t)(x->next() >> 32);
}

#define CK(expr)                                                                      \
    do {                                                                              \
        int rc_ = (expr);                                                             \
        if (rc_) {                                                                    \
            fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, bpg_last_error());          \
            g_failed++;                                                               \
            return false;                                                             \
        }                                                                             \
    } while (0)

static bool group(Xs &x, size_t K) {
    std::vector<bpg_prover *> ps(K, nullptr);
    std::vector<uint8_t> bl(32 * (size_t)g_nbits), V(32 * (size_t)g_nbits);
    std::vector<uint64_t> vars(g_nbits);
    for (size_t k = 0; k < K; k++) {
        for (size_t i = 0; i < g_nbits; i++) {                    // blindings below 2^248 < l
            for (int b = 0; b < 31; b++) bl[32 * i + b] = (uint8_t)(x.next() >> 24);
            bl[32 * i + 31] = 0;
        }
        CK(bpg_prover_new(g_gens, (const uint8_t *)"ScalarMul", 9, &ps[k]));
        CK(bpg_prover_commit_batch(ps[k], g_vals.data(), bl.data(), g_nbits, V.data(), vars.data()));
        bpg_cs *cs = bpg_prover_cs(ps[k]);
        bpg_point_gadget *b = nullptr, *e = nullptr, *r = nullptr;
        CK(bpg_point_from_coords(g_base.data(), &b));
        CK(bpg_point_from_coords(g_expected.data(), &e));
        CK(bpg_edwards_scalar_mul(cs, b, vars.data(), g_nbits, &r));
        CK(bpg_edwards_equal(cs, r, e));
        bpg_point_free(b);
        bpg_point_free(e);
        bpg_point_free(r);
    }
    std::vector<uint8_t> out(K << 16);
    std::vector<size_t> lens(K);
    CK(bpg_prover_prove_many(ps.data(), K, rng_cb, &x, out.data(), (size_t)1 << 16, lens.data()));
    for (auto *p : ps) bpg_prover_free(p);
    return true;
}

int main(int argc, char **argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: prove_driver STATEMENT THREADS GROUPS_PER_THREAD GROUP_SIZE\n");
        return 2;
    }
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(g_base.data(), 1, 128, f) != 128 || fread(g_expected.data(), 1, 128, f) != 128 || fread(&g_nbits, 4, 1, f) != 1) {
        fprintf(stderr, "cannot read %s\n", argv[1]);
        return 2;
    }
    g_vals.resize(32 * (size_t)g_nbits);
    if (fread(g_vals.data(), 1, g_vals.size(), f) != g_vals.size()) return 2;
    fclose(f);
    const int T = atoi(argv[2]), R = atoi(argv[3]);
    const size_t K = (size_t)atoi(argv[4]);
    bpg_ctx *ctx = nullptr;
    if (bpg_init(0, &ctx)) {
        fprintf(stderr, "bpg_init: %s\n", bpg_last_error());
        return 1;
    }
    size_t n = 27 * (size_t)g_nbits + 4, cap = 1;
    while (cap < n) cap <<= 1;
    if (bpg_gens_create(ctx, cap, 1, &g_gens)) {
        fprintf(stderr, "gens: %s\n", bpg_last_error());
        return 1;
    }
    std::mutex mu;
    std::condition_variable cv;
    int warm = 0;
    bool go = false;
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
            Xs x{0x9e3779b97f4a7c15ull * (uint64_t)(t + 1)};
            group(x, K);                                          // warm-up: the lane's buffers reach their size
            {
                std::unique_lock<std::mutex> lk(mu);
                warm++;
                cv.notify_all();
                cv.wait(lk, [&] { return go; });
            }
            for (int r = 0; r < R; r++)
                if (!group(x, K)) return;
        });
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return warm == T; });
    }
    struct rusage r0, r1;
    getrusage(RUSAGE_SELF, &r0);
    const auto t0 = std::chrono::steady_clock::now();
    {
        std::lock_guard<std::mutex> lk(mu);
        go = true;
    }
    cv.notify_all();
    for (auto &t : th) t.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    getrusage(RUSAGE_SELF, &r1);
    auto secs = [](const timeval &a) { return a.tv_sec + a.tv_usec * 1e-6; };
    const double cpu = secs(r1.ru_utime) - secs(r0.ru_utime) + secs(r1.ru_stime) - secs(r0.ru_stime);
    const double proofs = (double)T * R * (double)K;
    printf("{\"driver\": \"native\", \"threads\": %d, \"groups_per_thread\": %d, \"group\": %zu, \"multipliers\": %zu, \"proofs\": %.0f, "
           "\"seconds\": %.3f, \"proofs_per_s\": %.1f, \"host_cpu_ms_per_proof\": %.2f, \"failed_groups\": %d}\n",
           T, R, K, n, proofs, dt, proofs / dt, cpu * 1e3 / proofs, g_failed.load());
    bpg_gens_free(g_gens);
    bpg_shutdown(ctx);
    return g_failed.load() ? 1 : 0;
}
