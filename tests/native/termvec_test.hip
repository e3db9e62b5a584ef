// tests/native/termvec_test.hip -- TEST-ONLY host program (no HIP runtime call): exercises the small-vector
// behind LinearCombination (r1cs.h: TermVec) under AddressSanitizer / UBSan: growth past the inline
// capacity, copies, moves, self-append, assignment over heap and inline states.
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

static LC make(uint32_t start, uint32_t n) {
    LC lc;
    for (uint32_t i = 0; i < n; i++) lc.terms.push_back({Variable{VAR_LEFT, start + i}, Scalar(start + i + 1)});
    return lc;
}
static bool is_range(const LC &lc, uint32_t start, uint32_t n, size_t at = 0) {
    if (lc.terms.size() < at + n) return false;
    for (uint32_t i = 0; i < n; i++) {
        const Term &t = lc.terms[at + i];
        if (t.var.kind != VAR_LEFT || t.var.index != start + i || !(t.coeff == Scalar(start + i + 1))) return false;
    }
    return true;
}

int main() {
    for (uint32_t n : {0u, 1u, 5u, 6u, 7u, 13u, 100u}) {
        LC a = make(10, n);
        CHECK(a.terms.size() == n && is_range(a, 10, n));
        LC b = a;                                   // copy
        CHECK(b.terms.size() == n && is_range(b, 10, n));
        LC c = std::move(b);                        // move
        CHECK(c.terms.size() == n && is_range(c, 10, n) && b.terms.size() == 0);
        b = c;                                      // copy-assign into moved-from
        CHECK(is_range(b, 10, n) && b.terms.size() == n);
        c = make(500, 3);                           // move-assign over inline / heap
        CHECK(c.terms.size() == 3 && is_range(c, 500, 3));
        a += a;                                     // self-append
        CHECK(a.terms.size() == 2 * n && is_range(a, 10, n) && is_range(a, 10, n, n));
        LC d = a + c;                               // operator+ copies the left side
        CHECK(d.terms.size() == 2 * n + 3 && is_range(d, 500, 3, 2 * n));
        LC e = d - make(7, 2);                      // negated coefficients appended
        CHECK(e.terms.size() == 2 * n + 5);
        CHECK(e.terms[2 * n + 3].var.index == 7 && (e.terms[2 * n + 3].coeff == -Scalar(8)));
        std::vector<LC> v;                          // reallocation of a vector of LCs (moves)
        for (int k = 0; k < 20; k++) v.push_back(make(k, n));
        for (int k = 0; k < 20; k++) CHECK(is_range(v[k], k, n) && v[k].terms.size() == n);
        LC neg = -a;
        CHECK(neg.terms.size() == 2 * n);
        a = a;                                      // self-assignment
        CHECK(a.terms.size() == 2 * n);
    }
    printf(fails ? "termvec: %d failures\n" : "termvec: ok\n", fails);
    return fails ? 1 : 0;
}
