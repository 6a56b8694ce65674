/* Host-side Goldilocks vector arithmetic for INPUT SYNTHESIS only (era_boojum_amd/synthetic.py builds the satisfiable
 * SHA-shaped circuits the benches and tests prove).  Not on the proving path and not part of libboojum_hip.so: the
 * numpy implementation in field_np.py computes the same thing, this one is ~50x faster so that a 2^22-row circuit is
 * generated in seconds.  Built by era_boojum_amd/build.py with gcc -O3 -fopenmp. */
#include <omp.h>
#include <stddef.h>
#include <stdint.h>

/* cap the OpenMP team: the default (one thread per host core, 256 on the GPU boxes) costs more in fork/join than the
 * loops below take */
void synth_set_threads(int n) { omp_set_num_threads(n < 1 ? 1 : n); }

#define GL_P 0xFFFFFFFF00000001ULL

static inline uint64_t gl_canon(uint64_t a) { return a >= GL_P ? a - GL_P : a; }

static inline uint64_t gl_mul(uint64_t a, uint64_t b) {
    unsigned __int128 w = (unsigned __int128)a * b;
    uint64_t lo = (uint64_t)w, hi = (uint64_t)(w >> 64);
    uint64_t hh = hi >> 32, hl = hi & 0xFFFFFFFFULL;
    /* 2^64 = 2^32 - 1, 2^96 = -1 (mod p) */
    uint64_t t = lo - hh;
    if (lo < hh) t -= 0xFFFFFFFFULL;
    uint64_t m = hl * 0xFFFFFFFFULL;
    uint64_t r = t + m;
    if (r < m) r += 0xFFFFFFFFULL;
    return gl_canon(r);
}

static inline uint64_t gl_add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    if (s < a) s += 0xFFFFFFFFULL;
    return gl_canon(s);
}

void synth_mul(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n) {
#pragma omp parallel for schedule(static) if (n > 65536)
    for (size_t i = 0; i < n; i++) out[i] = gl_mul(gl_canon(a[i]), gl_canon(b[i]));
}

void synth_mul_scalar(const uint64_t *a, uint64_t s, uint64_t *out, size_t n) {
    s = gl_canon(s);
#pragma omp parallel for schedule(static) if (n > 65536)
    for (size_t i = 0; i < n; i++) out[i] = gl_mul(gl_canon(a[i]), s);
}

/* out = a*b + c*d */
void synth_fma2(const uint64_t *a, const uint64_t *b, const uint64_t *c, const uint64_t *d, uint64_t *out, size_t n) {
#pragma omp parallel for schedule(static) if (n > 65536)
    for (size_t i = 0; i < n; i++)
        out[i] = gl_add(gl_mul(gl_canon(a[i]), gl_canon(b[i])), gl_mul(gl_canon(c[i]), gl_canon(d[i])));
}

/* out[i] = base^(start + i) */
void synth_powers(uint64_t base, uint64_t *out, size_t n) {
    base = gl_canon(base);
    if (!n) return;
    const size_t B = 4096;
#pragma omp parallel for schedule(static)
    for (size_t blk = 0; blk < (n + B - 1) / B; blk++) {
        size_t i0 = blk * B, e = i0;
        uint64_t cur = 1, sq = base;
        while (e) {
            if (e & 1) cur = gl_mul(cur, sq);
            sq = gl_mul(sq, sq);
            e >>= 1;
        }
        for (size_t i = i0; i < i0 + B && i < n; i++) {
            out[i] = cur;
            cur = gl_mul(cur, base);
        }
    }
}
