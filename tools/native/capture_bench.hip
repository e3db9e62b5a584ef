// tools/native/capture_bench.hip -- host-only timing of circuit capture (no GPU call): the reference's
// SonnyEdwardsPointGadget::scalar_mul (ref: src/gadgets/point/edwards_point.rs:140-163) over `nbits` bit wires against
// a Prover, as BASELINE configs[2] captures it (2427 bits -> 65 533 multipliers).  The bits are allocated wires
// here (Prover::commit needs the GPU); the gadget's work is the same.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude tools/native/capture_bench.hip -o build/capture_bench \
//        -Lbulletproofs_gadgets_amd -lbpg -Wl,-rpath,$PWD/bulletproofs_gadgets_amd
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "../../bulletproofs_gadgets_amd/csrc/gadgets.h"

using namespace bpg;

static double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv) {
    const size_t nbits = argc > 1 ? (size_t)atol(argv[1]) : 2427;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int follow = argc > 3 ? atoi(argv[3]) : 1;
    SonnyPoint base{Scalar(0), Scalar(1), Scalar(1), Scalar(0)};
    // a point of the curve other than the identity: 8/9-ish is not needed for timing; use a few doublings of (x, y) = witness values
    base = SonnyPoint{Scalar(3), Scalar(5), Scalar(1), Scalar(15)};
#ifndef BASELINE
    std::shared_ptr<const CircuitKey> key;
#endif
    double best = 1e30, first = 0;
    size_t mults = 0, terms = 0;
    for (int r = 0; r < reps; r++) {
        Prover p(nullptr, "bench", 5);
#ifndef BASELINE
        if (follow && key) p.follow(key, nullptr);
#endif
        std::vector<Variable> sk(nbits);
        Scalar one(1), zero(0);
        for (size_t i = 0; i < nbits; i++) sk[i] = p.allocate_multiplier((i * 7 + r) % 3 ? &one : &zero, &zero).l;
        const double t0 = now_ms();
        auto q = SonnyEdwardsPointGadget::scalar_mul(SonnyEdwardsPointGadget::from_point(base), sk, p);
        q.equal(SonnyEdwardsPointGadget::from_point(base), p);
        const double dt = now_ms() - t0;
        if (r == 0) first = dt;
        if (dt < best) best = dt;
        mults = p.num_multipliers();
#ifndef BASELINE
        terms = p.store().nterms();
        if (!key) key = p.store().snapshot(p.num_multipliers());
        if (r == reps - 1) printf("followed_fully=%d\n", (int)p.store().followed_fully());
#else
        terms = p.store().term_var.size();
#endif
    }
    printf("nbits %zu multipliers %zu terms %zu: first %.2f ms, best %.2f ms (%.1f ns per multiplier)\n", nbits, mults, terms, first, best,
           best * 1e6 / (double)mults);
    return 0;
}
