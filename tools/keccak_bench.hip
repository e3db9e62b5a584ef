#include <chrono>
#include <cstdio>
#include "../bulletproofs_gadgets_amd/csrc/merlin.h"
using namespace bpg;
int main() {
    uint64_t s[25] = {1, 2, 3};
    alignas(64) uint64_t x[25][8] = {{1, 2, 3, 4, 5, 6, 7, 8}};
    const int N = 2000000;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) keccak_f1600(s);
    auto t1 = std::chrono::steady_clock::now();
    for (int i = 0; i < N / 8; i++) keccak_f1600_x8(x);
    auto t2 = std::chrono::steady_clock::now();
    double a = std::chrono::duration<double, std::nano>(t1 - t0).count() / N;
    double b = std::chrono::duration<double, std::nano>(t2 - t1).count() / N;
    printf("{\"keccak_f_ns\": %.1f, \"keccak_x8_ns_per_state\": %.1f, \"chk\": %llu}\n", a, b, (unsigned long long)(s[0] ^ x[0][0]));
    return 0;
}
