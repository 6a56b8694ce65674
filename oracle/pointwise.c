/* ORACLE — TEST INFRASTRUCTURE ONLY (see gl.h).
 * Pointwise helpers restated from cs/implementations/utils.rs:
 *   batch_inverse                      utils.rs:405-470   (Montgomery trick; result = elementwise inverse)
 *   batch_inverse_inplace_in_extension utils.rs:472-634
 */
#include "oracle.h"
#include <stdlib.h>

void orc_batch_inverse(const uint64_t *in, uint64_t *out, size_t n) {
    if (!n) return;
    gl_t *pre = (gl_t *)malloc(n * sizeof(gl_t));
    gl_t acc = 1;
    for (size_t i = 0; i < n; i++) { pre[i] = acc; acc = gl_mul(acc, gl_canon(in[i])); }
    gl_t inv = gl_inv(acc);
    for (size_t i = n; i-- > 0;) { gl_t x = gl_canon(in[i]); out[i] = gl_mul(inv, pre[i]); inv = gl_mul(inv, x); }
    free(pre);
}
void orc_ext_batch_inverse(const uint64_t *in0, const uint64_t *in1, uint64_t *o0, uint64_t *o1, size_t n) {
    for (size_t i = 0; i < n; i++) {
        gl2_t r = gl2_inv(gl2_make(gl_canon(in0[i]), gl_canon(in1[i])));
        o0[i] = r.c0; o1[i] = r.c1;
    }
}

/* ---- barycentric evaluation on a coset, restated from cs/implementations/utils.rs ----
 *   precompute_for_barycentric_evaluation_in_extension   utils.rs:907-1021
 *   barycentric_evaluate_base_at_extension_for_bitreversed_parallel / _extension_at_extension_   utils.rs:1085-1242
 * weights (stored bit-reversed): w_i = [coset*(z^n - coset^n)/(n*coset^n)] * omega^i / (z - coset*omega^i) */
#include <string.h>
void orc_barycentric_weights(unsigned log_n, uint64_t coset_in, const uint64_t *at, uint64_t *w0, uint64_t *w1) {
    size_t n = (size_t)1 << log_n;
    if (n == 1) { w0[0] = 1; w1[0] = 0; return; }
    gl_t coset = gl_canon(coset_in);
    gl2_t z = gl2_make(gl_canon(at[0]), gl_canon(at[1]));
    gl_t t = gl_pow(coset, n);
    gl2_t cf = gl2_pow(z, n);
    cf.c0 = gl_sub(cf.c0, t);
    cf = gl2_mul_base(cf, coset);
    cf = gl2_mul_base(cf, gl_inv(gl_mul(t, gl_from_u64((uint64_t)n))));
    gl_t omega = gl_omega(log_n), cur = coset, wpow = 1;
    for (size_t i = 0; i < n; i++) {
        gl2_t den = gl2_make(gl_sub(z.c0, cur), z.c1);
        gl2_t r = gl2_mul(gl2_inv(den), gl2_mul_base(cf, wpow));
        size_t j = bitrev64(i, log_n);
        w0[j] = r.c0; w1[j] = r.c1;
        cur = gl_mul(cur, omega); wpow = gl_mul(wpow, omega);
    }
}
/* f(z) for a base-field column given in the same (bit-reversed) order as the weights */
void orc_barycentric_eval_base(const uint64_t *values, const uint64_t *w0, const uint64_t *w1, size_t n, uint64_t *out2) {
    if (n == 1) { out2[0] = gl_canon(values[0]); out2[1] = 0; return; }
    gl_t a0 = 0, a1 = 0;
    for (size_t i = 0; i < n; i++) {
        gl_t f = gl_canon(values[i]);
        a0 = gl_add(a0, gl_mul(f, w0[i])); a1 = gl_add(a1, gl_mul(f, w1[i]));
    }
    out2[0] = a0; out2[1] = a1;
}
void orc_barycentric_eval_ext(const uint64_t *v0, const uint64_t *v1, const uint64_t *w0, const uint64_t *w1, size_t n, uint64_t *out2) {
    if (n == 1) { out2[0] = gl_canon(v0[0]); out2[1] = gl_canon(v1[0]); return; }
    gl2_t acc = gl2_make(0, 0);
    for (size_t i = 0; i < n; i++)
        acc = gl2_add(acc, gl2_mul(gl2_make(gl_canon(v0[i]), gl_canon(v1[i])), gl2_make(w0[i], w1[i])));
    out2[0] = acc.c0; out2[1] = acc.c1;
}

/* ---- DEEP quotient accumulation, restated from quotening_operation_in_extension (prover.rs:2523-2706) ----
 * dst(x) += [ sum_k ch_k * (f_k(x) - v_k) ] / (x - at)   over the LDE domain x_I = g * w_{nL}^{bitrev(I)}, I = coset*n + i.
 * src_c1[k] == NULL  <=>  f_k is a base-field polynomial (embedded as (f, 0)). */
/* one LDE point: out = [ sum_k ch_k * (f_k - v_k) ] / (x - at); f given as (f0[k], f1[k]) with is_ext[k] telling whether
 * source k has a c1 part.  This is the function the golden-proof test pins (tests/test_oracle_fixture.py). */
void orc_deep_quotient_point(const uint64_t *f0, const uint64_t *f1, const unsigned char *is_ext, size_t n_src,
                             const uint64_t *values, const uint64_t *challenges, const uint64_t *at, uint64_t x_in,
                             uint64_t *out2) {
    gl2_t z = gl2_make(gl_canon(at[0]), gl_canon(at[1]));
    gl_t x = gl_canon(x_in);
    gl2_t den = gl2_inv(gl2_make(gl_sub(x, z.c0), gl_neg(z.c1)));
    gl2_t acc = gl2_make(0, 0);
    for (size_t k = 0; k < n_src; k++) {
        gl2_t ch = gl2_make(gl_canon(challenges[2 * k]), gl_canon(challenges[2 * k + 1]));
        gl2_t v = gl2_make(gl_canon(values[2 * k]), gl_canon(values[2 * k + 1]));
        gl2_t f = gl2_make(gl_canon(f0[k]), is_ext[k] ? gl_canon(f1[k]) : 0);
        acc = gl2_add(acc, gl2_mul(gl2_sub(f, v), ch));
    }
    acc = gl2_mul(acc, den);
    out2[0] = acc.c0; out2[1] = acc.c1;
}

/* The same over the flat LDE indices [first, first + count) only: every array (sources, dst) holds `count` entries, entry i
 * belonging to the point I = first + i.  The accumulation is pointwise (prover.rs:2552-2706 iterates (outer, inner) and touches
 * one LDE point at a time), so a caller short of memory runs it one coset at a time (oracle/prover_streaming.py). */
void orc_deep_quotient_accumulate_range(const uint64_t *const *src_c0, const uint64_t *const *src_c1, size_t n_src,
                                        const uint64_t *values /*[n_src][2]*/, const uint64_t *challenges /*[n_src][2]*/,
                                        const uint64_t *at, unsigned log_n, unsigned log_lde, size_t first, size_t count,
                                        uint64_t *dst0, uint64_t *dst1, int threads) {
    unsigned log_full = log_n + log_lde;
    gl_t w = gl_omega(log_full);
#pragma omp parallel num_threads(threads)
    {
        uint64_t *f0 = (uint64_t *)malloc(n_src * sizeof(uint64_t)), *f1 = (uint64_t *)malloc(n_src * sizeof(uint64_t));
        unsigned char *ie = (unsigned char *)malloc(n_src);
        for (size_t k = 0; k < n_src; k++) ie[k] = src_c1[k] != NULL;
#pragma omp for schedule(static)
        for (size_t i = 0; i < count; i++) {
            gl_t x = gl_mul(GL_GEN, gl_pow(w, bitrev64(first + i, log_full)));
            for (size_t k = 0; k < n_src; k++) { f0[k] = src_c0[k][i]; f1[k] = src_c1[k] ? src_c1[k][i] : 0; }
            uint64_t o[2];
            orc_deep_quotient_point(f0, f1, ie, n_src, values, challenges, at, x, o);
            dst0[i] = gl_add(gl_canon(dst0[i]), o[0]);
            dst1[i] = gl_add(gl_canon(dst1[i]), o[1]);
        }
        free(f0); free(f1); free(ie);
    }
}

void orc_deep_quotient_accumulate(const uint64_t *const *src_c0, const uint64_t *const *src_c1, size_t n_src,
                                  const uint64_t *values /*[n_src][2]*/, const uint64_t *challenges /*[n_src][2]*/,
                                  const uint64_t *at, unsigned log_n, unsigned log_lde, uint64_t *dst0, uint64_t *dst1,
                                  int threads) {
    unsigned log_full = log_n + log_lde;
    size_t N = (size_t)1 << log_full;
    gl_t w = gl_omega(log_full);
#pragma omp parallel num_threads(threads)
    {
        uint64_t *f0 = (uint64_t *)malloc(n_src * sizeof(uint64_t)), *f1 = (uint64_t *)malloc(n_src * sizeof(uint64_t));
        unsigned char *ie = (unsigned char *)malloc(n_src);
        for (size_t k = 0; k < n_src; k++) ie[k] = src_c1[k] != NULL;
#pragma omp for schedule(static)
        for (size_t I = 0; I < N; I++) {
            gl_t x = gl_mul(GL_GEN, gl_pow(w, bitrev64(I, log_full)));
            for (size_t k = 0; k < n_src; k++) { f0[k] = src_c0[k][I]; f1[k] = src_c1[k] ? src_c1[k][I] : 0; }
            uint64_t o[2];
            orc_deep_quotient_point(f0, f1, ie, n_src, values, challenges, at, x, o);
            dst0[I] = gl_add(gl_canon(dst0[I]), o[0]);
            dst1[I] = gl_add(gl_canon(dst1[I]), o[1]);
        }
        free(f0); free(f1); free(ie);
    }
}
