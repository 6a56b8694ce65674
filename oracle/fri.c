/* ORACLE — TEST INFRASTRUCTURE ONLY (see gl.h).
 * FRI over F_p^2 (stored as two base columns c0, c1 in bit-reversed order), restated from
 *   compute_fri_schedule             cs/implementations/prover.rs:2281-2372
 *   do_fri                           cs/implementations/fri/mod.rs:49-358
 *   fold_multiple                    cs/implementations/fri/mod.rs:362-474
 *   interpolate_independent_cosets   cs/implementations/fri/mod.rs:476-585
 *   interpolate_flattened_cosets     cs/implementations/fri/mod.rs:587-678
 * Fold formula pinned by the golden proof.json (6-layer chain + final-monomial Horner check).
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#include <assert.h>

static unsigned ctz_sz(size_t x) { unsigned r = 0; while (!(x & 1)) { x >>= 1; r++; } return r; }

size_t orc_fri_schedule(uint32_t security_bits, size_t cap_size, uint32_t pow_bits, uint32_t rate_log2,
                        uint32_t initial_degree_log2, uint32_t *new_pow_bits_out, size_t *num_queries_out,
                        uint32_t *out_sched, size_t *final_degree) {
    uint32_t raw = security_bits - pow_bits, new_pow = pow_bits;
    if (raw % rate_log2 != 0) {
        if (new_pow >= rate_log2 - (raw % rate_log2)) new_pow -= rate_log2 - (raw % rate_log2);
    }
    raw = security_bits - new_pow;
    uint32_t nq = raw / rate_log2;
    if (raw % rate_log2 != 0) nq += 1;
    size_t cand = cap_size >> rate_log2;
    size_t stop = cand > 1 ? cand : 1;
    uint32_t stop_log = ctz_sz(stop), cap_log = ctz_sz(cap_size);
    uint32_t deg = initial_degree_log2;
    size_t len = 0;
    while (deg > stop_log) {
        if (deg + rate_log2 <= cap_log) break;
        if (deg - stop_log >= 3) { deg -= 3; out_sched[len++] = 3; }
        else if (deg - stop_log == 2) { deg -= 2; out_sched[len++] = 2; }
        else { deg -= 1; out_sched[len++] = 1; break; }
        if (deg + rate_log2 <= cap_log) break;
    }
    if (new_pow_bits_out) *new_pow_bits_out = new_pow;
    if (num_queries_out) *num_queries_out = nq;
    if (final_degree) *final_degree = (size_t)1 << deg;
    return len;
}

/* out[i] = (a + b) + alpha * ((a - b) * roots[i] * coset_inv), a = in[2i], b = in[2i+1]; no 1/2 factor */
void orc_fri_fold(const uint64_t *c0, const uint64_t *c1, size_t len, uint64_t *o0, uint64_t *o1,
                  const uint64_t *roots, uint64_t coset_inv, uint64_t ch0, uint64_t ch1) {
    gl2_t alpha = gl2_make(gl_canon(ch0), gl_canon(ch1));
    gl_t kappa = gl_canon(coset_inv);
    for (size_t i = 0; i < len / 2; i++) {
        gl_t a0 = gl_canon(c0[2 * i]), b0 = gl_canon(c0[2 * i + 1]);
        gl_t a1 = gl_canon(c1[2 * i]), b1 = gl_canon(c1[2 * i + 1]);
        gl_t r = gl_mul(gl_canon(roots[i]), kappa);
        gl2_t diff = gl2_make(gl_mul(gl_sub(a0, b0), r), gl_mul(gl_sub(a1, b1), r));
        gl2_t m = gl2_mul(diff, alpha);
        o0[i] = gl_add(gl_add(m.c0, a0), b0);
        o1[i] = gl_add(gl_add(m.c1, a1), b1);
    }
}

orc_fri_result *orc_do_fri(const uint64_t *c0, const uint64_t *c1, unsigned log_full, unsigned log_lde,
                           const uint32_t *schedule, size_t sched_len, size_t cap_size,
                           orc_transcript *t, int threads) {
    size_t full = (size_t)1 << log_full;
    orc_fri_result *r = (orc_fri_result *)calloc(1, sizeof(*r));
    r->num_oracles = sched_len;
    gl_t *roots = (gl_t *)malloc((full / 2) * sizeof(gl_t));
    orc_twiddles(roots, log_full, 1);                       /* fri/mod.rs:192-193 */
    gl_t kappa = gl_inv(GL_GEN);                            /* fri/mod.rs:195 */
    const uint64_t *cur0 = c0, *cur1 = c1;
    size_t cur_len = full;
    for (size_t step = 0; step < sched_len; step++) {
        unsigned k = schedule[step];
        /* oracle over the current array, 2^k values of c0 then c1 per leaf (merkle_tree.rs:176-386) */
        size_t E = (size_t)1 << k, leaves = cur_len / E;
        r->elems_per_leaf[step] = E; r->tree_leaves[step] = leaves;
        r->src_c0[step] = (uint64_t *)cur0; r->src_c1[step] = (uint64_t *)cur1; r->src_len[step] = cur_len;
        size_t cap = cap_size;
        r->trees[step] = (uint64_t *)malloc(4 * orc_merkle_tree_digests(leaves, cap) * sizeof(uint64_t));
        const uint64_t *srcs[2] = {cur0, cur1};
        orc_merkle_construct_chunked(srcs, 2, cur_len, E, cap, r->trees[step], threads);
        uint64_t capbuf[4 * 1024];
        orc_merkle_cap(r->trees[step], leaves, cap, capbuf);
        orc_transcript_absorb(t, capbuf, 4 * cap);
        gl_t ch0 = orc_transcript_challenge(t), ch1 = orc_transcript_challenge(t);
        r->challenges[step][0] = ch0; r->challenges[step][1] = ch1;
        gl2_t alpha = gl2_make(ch0, ch1);
        for (unsigned f = 0; f < k; f++) {
            size_t half = cur_len / 2;
            uint64_t *n0 = (uint64_t *)malloc(half * sizeof(uint64_t)), *n1 = (uint64_t *)malloc(half * sizeof(uint64_t));
            orc_fri_fold(cur0, cur1, cur_len, n0, n1, roots, kappa, alpha.c0, alpha.c1);
            kappa = gl_sqr(kappa);
            alpha = gl2_sqr(alpha);
            if (f > 0) { free((void *)cur0); free((void *)cur1); }   /* inner temporaries of this step */
            cur0 = n0; cur1 = n1; cur_len = half;
        }
    }
    /* final interpolation: fri/mod.rs:312-343 */
    unsigned log_m = ctz_sz(cur_len);
    uint64_t *f0 = (uint64_t *)malloc(cur_len * sizeof(uint64_t)), *f1 = (uint64_t *)malloc(cur_len * sizeof(uint64_t));
    memcpy(f0, cur0, cur_len * sizeof(uint64_t)); memcpy(f1, cur1, cur_len * sizeof(uint64_t));
    orc_bitreverse(f0, log_m); orc_bitreverse(f1, log_m);
    gl_t coset = gl_inv(kappa);
    orc_ifft_natural_to_natural(f0, log_m, coset, roots);    /* roots[..m/2] prefix property */
    orc_ifft_natural_to_natural(f1, log_m, coset, roots);
    r->final_degree = cur_len >> log_lde;
    r->final_c0 = f0; r->final_c1 = f1;
    /* keep the last folded array alive as "source of a would-be next oracle" for query tests */
    r->src_c0[sched_len] = (uint64_t *)cur0; r->src_c1[sched_len] = (uint64_t *)cur1; r->src_len[sched_len] = cur_len;
    orc_transcript_absorb(t, f0, r->final_degree);
    orc_transcript_absorb(t, f1, r->final_degree);
    free(roots);
    return r;
}
void orc_fri_result_free(orc_fri_result *r) {
    if (!r) return;
    for (size_t i = 0; i < r->num_oracles; i++) free(r->trees[i]);
    for (size_t i = 1; i <= r->num_oracles; i++) { free(r->src_c0[i]); free(r->src_c1[i]); }
    free(r->final_c0); free(r->final_c1); free(r);
}
