/* ORACLE — TEST INFRASTRUCTURE ONLY (see gl.h).
 * Radix-2 NTT / iNTT / LDE over Goldilocks, restated from the reference:
 *   precompute_twiddles_for_fft          cs/implementations/utils.rs:88-125
 *   bitreverse_enumeration_inplace       fft/mod.rs:41-155
 *   distribute_powers                    fft/mod.rs:308-317
 *   fft_natural_to_bitreversed           fft/mod.rs:398-411
 *   ifft_natural_to_natural              fft/mod.rs:464-491
 *   serial_ct_ntt_natural_to_bitreversed fft/mod.rs:659-734
 *   transform_monomials_to_lde           cs/implementations/utils.rs:311-403
 */
#include "oracle.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ntt_avx512.c: the butterfly loops eight at a time, where the CPU has AVX-512 (run-time check shared with the Poseidon2 path) */
void orc_ct_bfly_x8(uint64_t *lo, uint64_t *hi, size_t count, uint64_t s);
void orc_ct_addsub_x8(uint64_t *lo, uint64_t *hi, size_t count);
void orc_scale_x8(uint64_t *a, size_t count, uint64_t s);
void orc_distribute_powers_x8(uint64_t *a, size_t count, uint64_t el);
void orc_canonicalize_x8(uint64_t *a, size_t count);
static inline int vec(void) { return orc_poseidon2_avx512_available(); }

/* T[j] = w^{bitrev_{log_n-1}(j)}, j < n/2; w = omega_n or its inverse (utils.rs:88-125) */
void orc_twiddles(uint64_t *out, unsigned log_n, int inverse) {
    if (log_n == 0) return;
    size_t half = (size_t)1 << (log_n - 1);
    gl_t w = gl_omega(log_n);
    if (inverse) w = gl_inv(w);
    gl_t *nat = (gl_t *)malloc(half * sizeof(gl_t));
    gl_t cur = 1;
    for (size_t i = 0; i < half; i++) { nat[i] = cur; cur = gl_mul(cur, w); }
    for (size_t i = 0; i < half; i++) out[i] = nat[bitrev64(i, log_n - 1)];
    free(nat);
}

/* The batched drivers below ask for the same table again and again (the coset-streaming restatement: ~80 calls at one size, each
 * n / 2 chained products and a scattered pass): the largest table per direction is kept — T for 2^k is a prefix of T for 2^(k+1),
 * bitrev_k(j) = 2 bitrev_(k-1)(j) for j < 2^(k-1).  Tables that were outgrown are not freed (another thread may still read them). */
static const gl_t *cached_twiddles(unsigned log_n, int inverse) {
    static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    static gl_t *tab[2];
    static unsigned have[2];
    inverse = inverse ? 1 : 0;
    pthread_mutex_lock(&mu);
    if (!tab[inverse] || have[inverse] < log_n) {
        gl_t *t = (gl_t *)malloc((((size_t)1 << log_n) / 2 + 1) * sizeof(gl_t));
        orc_twiddles(t, log_n, inverse);
        tab[inverse] = t;
        have[inverse] = log_n;
    }
    const gl_t *r = tab[inverse];
    pthread_mutex_unlock(&mu);
    return r;
}

void orc_bitreverse(uint64_t *a, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; i++) {
        size_t j = bitrev64(i, log_n);
        if (i < j) { uint64_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
}

void orc_canonicalize(uint64_t *a, size_t n) {
    size_t i = 0;
    if (vec()) { i = n / 8 * 8; orc_canonicalize_x8(a, i); }
    for (; i < n; i++) a[i] = gl_canon(a[i]);
}

/* fft/mod.rs:308-317 */
static void distribute_powers(gl_t *a, size_t n, gl_t el) {
    if (vec() && n >= 8) { orc_distribute_powers_x8(a, n, el); return; }   /* n is a power of two */
    gl_t s = 1;
    for (size_t i = 0; i < n; i++) { a[i] = gl_mul(a[i], s); s = gl_mul(s, el); }
}

/* fft/mod.rs:659-734 : round r has 2^r groups, group k uses tw[k]; butterfly (u, v*s) -> (u+v*s, u-v*s).
 * The rounds are the reference's; their ORDER OF EXECUTION is blocked for the cache the way the reference's "cache friendly"
 * variant does it (fft/mod.rs:504 bench): once a group is no larger than CT_BLOCK elements, all remaining rounds of that group
 * run back to back while it sits in L2 (group k of round r splits into groups 2k, 2k+1 of round r+1, so the sub-transform
 * of a block only ever touches the block).  Same butterflies, same results. */
#define CT_BLOCK ((size_t)1 << 15)
static void ct_rounds_in_block(gl_t *a, size_t base, size_t len, size_t first_group, const gl_t *tw) {
    /* the block [base, base + len) is group `first_group` of its round: run this and all later rounds inside it */
    size_t groups = 1, pairs = len / 2;
    while (pairs >= 1) {
        for (size_t g = 0; g < groups; g++) {
            size_t i1 = base + g * pairs * 2, i2 = i1 + pairs;
            gl_t s = tw[first_group * groups + g];
            if (pairs >= 8 && vec()) { orc_ct_bfly_x8(a + i1, a + i2, pairs, s); continue; }
            for (size_t j = i1; j < i2; j++) {
                gl_t u = a[j], v = gl_mul(a[j + pairs], s);
                a[j + pairs] = gl_sub(u, v);
                a[j] = gl_add(u, v);
            }
        }
        groups *= 2;
        pairs /= 2;
    }
}
static void serial_ct_ntt(gl_t *a, unsigned log_n, const gl_t *tw) {
    size_t n = (size_t)1 << log_n;
    if (n == 1) return;
    size_t pairs = n / 2, groups = 1, dist = n / 2;
    if (pairs >= 8 && vec()) orc_ct_addsub_x8(a, a + dist, pairs);
    else for (size_t j = 0; j < pairs; j++) {           /* omega = 1 special case */
        gl_t u = a[j], v = a[j + dist];
        a[j + dist] = gl_sub(u, v);
        a[j] = gl_add(u, v);
    }
    pairs /= 2; groups *= 2; dist /= 2;
    while (groups < n && 2 * pairs > CT_BLOCK) {   /* streaming rounds: groups larger than the cache block */
        for (size_t k = 0; k < groups; k++) {
            size_t i1 = k * pairs * 2, i2 = i1 + pairs;
            gl_t s = tw[k];
            if (pairs >= 8 && vec()) { orc_ct_bfly_x8(a + i1, a + i1 + dist, pairs, s); continue; }
            for (size_t j = i1; j < i2; j++) {
                gl_t u = a[j], v = gl_mul(a[j + dist], s);
                a[j + dist] = gl_sub(u, v);
                a[j] = gl_add(u, v);
            }
        }
        pairs /= 2; groups *= 2; dist /= 2;
    }
    if (groups < n)                                 /* every remaining group fits the block: finish them one by one */
        for (size_t k = 0; k < groups; k++) ct_rounds_in_block(a, k * pairs * 2, pairs * 2, k, tw);
}

/* fft/mod.rs:398-411 */
void orc_fft_natural_to_bitreversed(uint64_t *a, unsigned log_n, uint64_t coset, const uint64_t *tw) {
    size_t n = (size_t)1 << log_n;
    orc_canonicalize(a, n);
    coset = gl_canon(coset);
    if (coset != 1) distribute_powers(a, n, coset);
    serial_ct_ntt(a, log_n, tw);
}

/* fft/mod.rs:464-491 */
void orc_ifft_natural_to_natural(uint64_t *a, unsigned log_n, uint64_t coset, const uint64_t *inv_tw) {
    size_t n = (size_t)1 << log_n;
    orc_canonicalize(a, n);
    coset = gl_canon(coset);
    serial_ct_ntt(a, log_n, inv_tw);
    orc_bitreverse(a, log_n);
    if (coset != 1) distribute_powers(a, n, gl_inv(coset));
    if (n > 1) {
        gl_t n_inv = gl_inv(gl_from_u64((uint64_t)n));
        if (vec() && n >= 8) orc_scale_x8(a, n, n_inv);
        else for (size_t i = 0; i < n; i++) a[i] = gl_mul(a[i], n_inv);
    }
}

/* O(n^2) evaluation used by the reference's own differential tests (fft/mod.rs:1345-1384):
 * out[k] = sum_i a[i] * (coset*w^k)^i, natural order */
void orc_naive_dft(const uint64_t *a, uint64_t *out, unsigned log_n, uint64_t coset) {
    size_t n = (size_t)1 << log_n;
    gl_t w = gl_omega(log_n);
    coset = gl_canon(coset);
    for (size_t k = 0; k < n; k++) {
        gl_t x = gl_mul(coset, gl_pow(w, k)), acc = 0, xp = 1;
        for (size_t i = 0; i < n; i++) { acc = gl_add(acc, gl_mul(gl_canon(a[i]), xp)); xp = gl_mul(xp, x); }
        out[k] = acc;
    }
}

/* LDE coset shifts: shift_c = g * omega_{nL}^{bitrev_{log L}(c)}  (utils.rs:345-346, 370-373) */
void orc_lde_coset_shifts(uint64_t *out, unsigned log_n, unsigned log_lde) {
    size_t L = (size_t)1 << log_lde;
    gl_t w = gl_omega(log_n + log_lde);
    for (size_t c = 0; c < L; c++) out[c] = gl_mul(GL_GEN, gl_pow(w, bitrev64(c, log_lde)));
}

/* transform_monomials_to_lde for ONE column: mono[n] -> out[L][n], each coset bit-reversed (utils.rs:311-403) */
void orc_lde_from_monomials(const uint64_t *mono, uint64_t *out, unsigned log_n, unsigned log_lde, const uint64_t *fwd_tw) {
    size_t n = (size_t)1 << log_n, L = (size_t)1 << log_lde;
    gl_t shifts[64];
    orc_lde_coset_shifts(shifts, log_n, log_lde);
    for (size_t c = 0; c < L; c++) {
        memcpy(out + c * n, mono, n * sizeof(uint64_t));
        orc_fft_natural_to_bitreversed(out + c * n, log_n, shifts[c], fwd_tw);
    }
}

/* ---- batched drivers ("one polynomial per core", fft/mod.rs:284-287, utils.rs:295-304, 363-379).
 * These are what bench.py's cpu_baseline leg times. ---- */
void orc_fft_batch(uint64_t *cols, unsigned log_n, size_t n_cols, uint64_t coset, int threads) {
    size_t n = (size_t)1 << log_n;
    const gl_t *tw = cached_twiddles(log_n, 0);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (size_t c = 0; c < n_cols; c++) orc_fft_natural_to_bitreversed(cols + c * n, log_n, coset, tw);
}
/* out-of-place: dst[c] = NTT(src[c]); the copy runs inside the parallel loop (a caller that reuses dst over many cosets does not
 * fault fresh pages in for every call — oracle/prover_streaming.py) */
void orc_fft_batch_to(const uint64_t *src, uint64_t *dst, unsigned log_n, size_t n_cols, uint64_t coset, int threads) {
    size_t n = (size_t)1 << log_n;
    const gl_t *tw = cached_twiddles(log_n, 0);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (size_t c = 0; c < n_cols; c++) {
        memcpy(dst + c * n, src + c * n, n * sizeof(uint64_t));
        orc_fft_natural_to_bitreversed(dst + c * n, log_n, coset, tw);
    }
}
void orc_ifft_batch(uint64_t *cols, unsigned log_n, size_t n_cols, uint64_t coset, int threads) {
    size_t n = (size_t)1 << log_n;
    const gl_t *tw = cached_twiddles(log_n, 1);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (size_t c = 0; c < n_cols; c++) orc_ifft_natural_to_natural(cols + c * n, log_n, coset, tw);
}
/* mono [n_cols][n] -> out [n_cols][L][n] */
void orc_lde_batch(const uint64_t *mono, uint64_t *out, unsigned log_n, unsigned log_lde, size_t n_cols, int threads) {
    size_t n = (size_t)1 << log_n, L = (size_t)1 << log_lde;
    const gl_t *tw = cached_twiddles(log_n, 0);
    gl_t shifts[64];
    orc_lde_coset_shifts(shifts, log_n, log_lde);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (size_t job = 0; job < n_cols * L; job++) {
        size_t col = job / L, c = job % L;
        uint64_t *dst = out + (col * L + c) * n;
        memcpy(dst, mono + col * n, n * sizeof(uint64_t));
        orc_fft_natural_to_bitreversed(dst, log_n, shifts[c], tw);
    }
}
