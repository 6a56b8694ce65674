/* Finds operand pairs that drive gl::mul_weak (era_boojum_amd/csrc/gl.cuh) into its rare branches: the final subtraction
 * R - H.hi - cm borrows (probability ~2^-32 per random product), with and without the carry of the preceding multiply-add.
 * The model below is the instruction sequence of mul_weak on 64-bit words; every candidate is checked against a*b mod p
 * computed with 128-bit integers.  Output: JSON lines for tests/golden/gl_mul_rare.json.
 *   gcc -O3 -fopenmp tools/find_rare_mul_vectors.c -o /tmp/find_rare && /tmp/find_rare > tests/golden/gl_mul_rare.json */
#include <stdint.h>
#include <stdio.h>
#include <omp.h>
typedef unsigned __int128 u128;
static const uint64_t P = 0xFFFFFFFF00000001ull, EPS = 0xFFFFFFFFull;
static uint64_t splitmix(uint64_t *s) { uint64_t z = (*s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
/* returns the weak result; *cls: bit0 = borrow, bit1 = carry c, bit2 = cm */
static uint64_t model(uint64_t a, uint64_t b, int *cls) {
    uint32_t a0 = (uint32_t)a, a1 = a >> 32, b0 = (uint32_t)b, b1 = b >> 32;
    uint64_t T = (uint64_t)a0 * b0;
    uint64_t U = (uint64_t)a0 * b1 + (T >> 32);
    u128 Xw = (u128)((uint64_t)a1 * b0) + U;
    uint64_t X = (uint64_t)Xw; int cm = (int)(Xw >> 64);
    uint64_t H = (uint64_t)a1 * b1 + (X >> 32);
    uint64_t lo = (uint32_t)T | (X << 32);
    u128 Rw = (u128)((uint64_t)(uint32_t)H * EPS) + lo;
    uint64_t R = (uint64_t)Rw; int c = (int)(Rw >> 64);
    uint64_t sub = (H >> 32) + (uint64_t)cm;
    int bo = R < sub;
    uint64_t D = R - sub;
    if (bo && !c) D -= EPS;
    uint64_t V = D + (c ? EPS : 0);
    *cls = bo | (c << 1) | (cm << 2);
    return V;
}
int main(void) {
    int found[8] = {0};
    const int want = 6;
    /* structured seeds first: a = 2^48, b = k * 2^48 gives a * b = k * 2^96 = -k: borrow without carry */
    for (uint64_t k = 1; k <= 3; k++) {
        uint64_t a = 1ull << 48, b = k << 48; int cls; uint64_t v = model(a, b, &cls);
        uint64_t ref = (uint64_t)(((u128)a * b) % P);
        printf("{\"a\": %llu, \"b\": %llu, \"product\": %llu, \"class\": %d, \"ok\": %d}\n", (unsigned long long)a, (unsigned long long)b, (unsigned long long)ref, cls, (v % P) == ref);
    }
#pragma omp parallel
    {
        uint64_t seed = 0x1234567ull * (omp_get_thread_num() + 1);
        for (uint64_t it = 0; it < (1ull << 36); it++) {
            int done; 
#pragma omp atomic read
            done = found[3];
            if (done >= want) break;   /* borrow WITHOUT carry never shows up at random (2^-64): the structured seeds above cover it */
            /* bias the search: borrow needs R < 2^32, i.e. lo + H.lo*EPS == small (mod 2^64).  Pick a, b1 at random and solve for
               b0 by scanning is too slow; instead pick a random a and random b and test (2^-32), but 8 threads x ~4e8/s reach
               a hit in seconds only with help: force a0 = 0 so T = U = 0, X = a1*b0, lo = X.lo << 32, R = (X.lo + H.lo) << 32 - H.lo */
            uint64_t a = splitmix(&seed), b = splitmix(&seed);
            if (it & 1) a &= 0xFFFFFFFF00000000ull;
            int cls; uint64_t v = model(a, b, &cls);
            if (cls & 1) {
                uint64_t ref = (uint64_t)(((u128)a * b) % P);
#pragma omp critical
                {
                    int k = cls & 3;
                    if (found[k] < want) {
                        found[k]++;
                        printf("{\"a\": %llu, \"b\": %llu, \"product\": %llu, \"class\": %d, \"ok\": %d}\n", (unsigned long long)a, (unsigned long long)b, (unsigned long long)ref, cls, (v % P) == ref);
                        fflush(stdout);
                    }
                }
            }
        }
    }
    return 0;
}
