// tests/native/termvec_test.hip -- TEST-ONLY host program (no HIP runtime call): exercises the storage of
// LinearCombination (r1cs.h: packed terms, side table of large coefficients, heap spill) under AddressSanitizer /
// UBSan: growth past the inline capacities, copies, moves, self-append, assignment over heap and inline states,
// small and large coefficients and their negation.
// Build + run: tests/test_host_r1cs.py::test_termvec_under_sanitizers
#include <cstdio>
#include <utility>
#include <vector>

#include "../../bulletproofs_gadgets_amd/csrc/r1cs.h"

using namespace bpg;

static int fails = 0;
#define CHECK(c)                                              \
    do {                                                      \
        if (!(c)) {                                           \
            printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); \
            fails++;                                          \
        }                                                     \
    } while (0)

static Scalar coeff_of(uint32_t x) {
    Scalar c(x + 1);
    if (x % 3 == 2) c = c * Scalar(0x123456789abcdefULL) * Scalar(0xfedcba987654321ULL) * Scalar(0x1111111111111111ULL);
    return c;
}
static LC make(uint32_t start, uint32_t n) {
    LC lc;
    // every third coefficient is large (side table), the rest small integers
    for (uint32_t i = 0; i < n; i++) lc.push(Variable{VAR_LEFT, start + i}, coeff_of(start + i));
    return lc;
}
static bool is_range(const LC &lc, uint32_t start, uint32_t n, size_t at = 0) {
    if (lc.size() < at + n) return false;
    for (uint32_t i = 0; i < n; i++) {
        const Term t = lc.term(at + i);
        if (t.var.kind != VAR_LEFT || t.var.index != start + i || !(t.coeff == coeff_of(start + i))) return false;
    }
    return true;
}

int main() {
    for (uint32_t n : {0u, 1u, 5u, 6u, 7u, 13u, 100u}) {
        LC a = make(10, n);
        CHECK(a.size() == n && is_range(a, 10, n));
        LC b = a;                                   // copy
        CHECK(b.size() == n && is_range(b, 10, n));
        LC c = std::move(b);                        // move
        CHECK(c.size() == n && is_range(c, 10, n) && b.size() == 0);
        b = c;                                      // copy-assign into moved-from
        CHECK(is_range(b, 10, n) && b.size() == n);
        c = make(500, 3);                           // move-assign over inline / heap
        CHECK(c.size() == 3 && is_range(c, 500, 3));
        a += a;                                     // self-append
        CHECK(a.size() == 2 * n && is_range(a, 10, n) && is_range(a, 10, n, n));
        LC d = a + c;                               // operator+ copies the left side
        CHECK(d.size() == 2 * n + 3 && is_range(d, 500, 3, 2 * n));
        LC e = d - make(7, 2);                      // negated coefficients appended
        CHECK(e.size() == 2 * n + 5);
        CHECK(e.term(2 * n + 3).var.index == 7 && (e.term(2 * n + 3).coeff == -coeff_of(7)));
        CHECK(e.term(2 * n + 4).var.index == 8 && (e.term(2 * n + 4).coeff == -coeff_of(8)));      // a large coefficient, negated
        std::vector<LC> v;                          // reallocation of a vector of LCs (moves)
        for (int k = 0; k < 20; k++) v.push_back(make(k, n));
        for (int k = 0; k < 20; k++) CHECK(is_range(v[k], k, n) && v[k].size() == n);
        LC neg = -a;
        CHECK(neg.size() == 2 * n);
        for (size_t i = 0; i < neg.size(); i++) CHECK(neg.term(i).coeff == -a.term(i).coeff && neg.term(i).var.index == a.term(i).var.index);
        // small_of / scalar_of round trip, at the edges of the small range
        for (int32_t k : {0, 1, -1, 2, -2, (1 << 30) - 1, -((1 << 30) - 1)}) CHECK(LC::small_of(LC::scalar_of(k)) == k);
        CHECK(LC::small_of(Scalar((uint64_t)1 << 30).v) == LC::BIG && LC::small_of((-Scalar((uint64_t)1 << 30)).v) == LC::BIG);
        a = a;                                      // self-assignment
        CHECK(a.size() == 2 * n);
    }
    printf(fails ? "termvec: %d failures\n" : "termvec: ok\n", fails);
    return fails ? 1 : 0;
}
