// Host check of the weak-residue helpers of csrc/gl.h (the host halves of the __host__ __device__ functions share every line
// with the device code except gl::mul_weak, whose device form is inline assembly and is checked on the GPU through
// bj_field_op_batch): add_weak, sub_weak, mul7_weak, e2_mul_weak, mul_pow2 and the canonical operators on extreme words and
// random ones, against unsigned __int128 arithmetic.      hipcc --cuda-host-only -x hip tools/gl_weak_host_check.cpp
#include "gl.h"

#include <cstdio>
#include <vector>

typedef unsigned __int128 u128;
using gl::u64;
static const u64 P = gl::P;

static u64 red(u128 x) { return (u64)(x % P); }
static u64 splitmix(u64 &s) {
    u64 z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

int main() {
    std::vector<u64> w = {0, 1, 2, 7, 0xFFFFFFFFULL, 0x100000000ULL, 0xFFFFFFFF00000000ULL, P - 1, P, P + 1, P + 0xFFFFFFFEULL,
                          0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFF00000001ULL, 0x8000000000000000ULL,
                          0x7FFFFFFFFFFFFFFFULL, 0xFFFFFFFE00000002ULL, 0x00000000FFFFFFFEULL};
    u64 seed = 42;
    for (int i = 0; i < 4000; i++) w.push_back(splitmix(seed));
    for (int i = 0; i < 200; i++) w.push_back(0xFFFFFFFFFFFFFFFFULL - (splitmix(seed) & 0x1FFFFFFFFULL));   // within 2^33 of 2^64
    for (int i = 0; i < 200; i++) w.push_back(splitmix(seed) & 0x1FFFFFFFFULL);                              // within 2^33 of 0
    size_t bad = 0, checks = 0;
    auto expect = [&](const char *what, u64 got, u64 want_mod_p, u64 a, u64 b) {
        checks++;
        if (got % P != want_mod_p) {
            if (bad++ < 10) std::printf("%s(%016llx, %016llx): got %016llx want %016llx (mod p)\n", what, (unsigned long long)a,
                                        (unsigned long long)b, (unsigned long long)got, (unsigned long long)want_mod_p);
        }
    };
    const size_t n = w.size();
    for (size_t i = 0; i < n; i++) {
        const u64 a = w[i];
        expect("mul7_weak", gl::mul7_weak(a), red((u128)a * 7), a, 7);
        for (unsigned k = 0; k < 32; k += 5) expect("mul_pow2", gl::mul_pow2(gl::canon(a), k), red((u128)gl::canon(a) << k), a, k);
        for (size_t j = (i * 7) % 13; j < n; j += 13) {
            const u64 b = w[j];
            expect("add_weak", gl::add_weak(a, b), red((u128)a + b), a, b);
            expect("sub_weak", gl::sub_weak(a, b), red((u128)a + (u128)P * 2 - (u128)(b % P)), a, b);
            const u64 ca = gl::canon(a % P), cb = gl::canon(b % P);
            u64 r = gl::add(ca, cb);
            checks++;
            if (r != red((u128)ca + cb)) bad++;
            r = gl::sub(ca, cb);
            checks++;
            if (r != red((u128)ca + P - cb)) bad++;
            r = gl::mul(a, b);   // canonical result for ANY operands
            checks++;
            if (r != red((u128)(a % P) * (b % P))) { if (bad++ < 10) std::printf("mul(%016llx, %016llx) = %016llx\n", (unsigned long long)a, (unsigned long long)b, (unsigned long long)r); }
        }
    }
    // the F_p^2 product on weak words against the canonical Karatsuba form
    for (size_t i = 0; i + 3 < n; i += 3) {
        const gl::e2 a{w[i], w[i + 1]}, b{w[i + 2], w[i + 3]};
        const gl::e2 got = gl::e2_mul_weak(a, b);
        const gl::e2 want = gl::e2_mul(gl::e2{a.c0 % P, a.c1 % P}, gl::e2{b.c0 % P, b.c1 % P});
        checks += 2;
        if (got.c0 % P != want.c0 || got.c1 % P != want.c1) {
            if (bad++ < 10) std::printf("e2_mul_weak mismatch at %zu\n", i);
        }
        const u64 c0 = red((u128)(a.c0 % P) * (b.c0 % P) + (u128)7 * red((u128)(a.c1 % P) * (b.c1 % P)));
        checks++;
        if (want.c0 != c0) bad++;
    }
    std::printf("%zu checks, %zu mismatches\n", checks, bad);
    if (!bad) std::printf("weak helpers == 128-bit arithmetic\n");
    return bad ? 1 : 0;
}
