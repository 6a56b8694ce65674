/* ORACLE — TEST INFRASTRUCTURE ONLY (see gl.h).
 * Bulk per-stage operations of prove_cpu_basic rounds 2 and 3, restated from the reference:
 *   compute_partial_products_in_extension / pointwise_rational / shifted_grand_product
 *                                                    cs/implementations/copy_permutation.rs:114-248, 425-510, 649-830
 *   compute_lookup_poly_pairs_specialized            cs/implementations/lookup_argument_in_ext.rs:320-700
 *   gate evaluation over general purpose columns     cs/implementations/prover.rs:1031-1080,
 *                                                    cs/implementations/buffering_source.rs:133-222, 304-362,
 *                                                    cs/gates/{constant_allocator,fma_gate_without_constant,reduction_gate}.rs
 *   compute_selector_subpath                         cs/implementations/prover.rs:2775-2916
 *   (z-1)*L1 term, unnormalized_l1_inverse           cs/implementations/prover.rs:1189-1227, utils.rs:1585-1672
 *   compute_quotient_terms_in_extension              cs/implementations/copy_permutation.rs:1000-1249
 *   compute_quotient_terms_for_lookup_specialized    cs/implementations/lookup_argument_in_ext.rs:949-1319
 *   divide_by_vanishing_for_bitreversed_coset_enumeration   cs/implementations/utils.rs:770-817
 * All arrays are canonical u64; columns are contiguous [col][len].
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

/* z and the partial products over the main domain (natural row order).
 * vars/sigmas: [V][n]; out_z: [2][n]; out_partials: [n_chunks-1][2][n], n_chunks = ceil(V / chunk). */
void orc_copy_perm_stage2(const uint64_t *vars, const uint64_t *sigmas, const uint64_t *non_res, size_t V,
                          unsigned log_n, size_t chunk, const uint64_t *beta2, const uint64_t *gamma2,
                          uint64_t *out_z, uint64_t *out_partials, int threads) {
    size_t n = (size_t)1 << log_n;
    size_t n_chunks = (V + chunk - 1) / chunk;
    gl2_t beta = gl2_make(gl_canon(beta2[0]), gl_canon(beta2[1])), gamma = gl2_make(gl_canon(gamma2[0]), gl_canon(gamma2[1]));
    gl_t omega = gl_omega(log_n);
    /* P[j][row] = prod_{i in chunk j} (w + beta*k*x + gamma) / (w + beta*sigma + gamma) */
    gl2_t *Pj = (gl2_t *)malloc(n_chunks * n * sizeof(gl2_t));
    gl2_t *almost = (gl2_t *)malloc(n * sizeof(gl2_t));
#pragma omp parallel for schedule(static) num_threads(threads)
    for (size_t r = 0; r < n; r++) {
        gl_t x = gl_pow(omega, r);
        gl2_t total = gl2_make(1, 0);
        for (size_t j = 0; j < n_chunks; j++) {
            gl2_t num = gl2_make(1, 0), den = gl2_make(1, 0);
            for (size_t i = j * chunk; i < (j + 1) * chunk && i < V; i++) {
                gl_t w = gl_canon(vars[i * n + r]);
                gl_t kx = gl_mul(gl_canon(non_res[i]), x);
                gl2_t a = gl2_make(gl_add(gl_add(gl_mul(kx, beta.c0), w), gamma.c0), gl_add(gl_mul(kx, beta.c1), gamma.c1));
                gl_t s = gl_canon(sigmas[i * n + r]);
                gl2_t b = gl2_make(gl_add(gl_add(gl_mul(s, beta.c0), w), gamma.c0), gl_add(gl_mul(s, beta.c1), gamma.c1));
                num = gl2_mul(num, a);
                den = gl2_mul(den, b);
            }
            gl2_t p = gl2_mul(num, gl2_inv(den));
            Pj[j * n + r] = p;
            total = gl2_mul(total, p);
        }
        almost[r] = total;
    }
    /* shifted grand product: z[0] = 1, z[r] = prod_{r' < r} almost[r'] */
    gl2_t acc = gl2_make(1, 0);
    for (size_t r = 0; r < n; r++) {
        out_z[r] = acc.c0; out_z[n + r] = acc.c1;
        acc = gl2_mul(acc, almost[r]);
    }
    /* partial_0 = z * P_0, partial_k = partial_{k-1} * P_k, for k < n_chunks - 1 */
#pragma omp parallel for schedule(static) num_threads(threads)
    for (size_t r = 0; r < n; r++) {
        gl2_t cur = gl2_make(out_z[r], out_z[n + r]);
        for (size_t j = 0; j + 1 < n_chunks; j++) {
            cur = gl2_mul(cur, Pj[j * n + r]);
            out_partials[(2 * j) * n + r] = cur.c0;
            out_partials[(2 * j + 1) * n + r] = cur.c1;
        }
    }
    free(Pj); free(almost);
}

/* A_i = 1 / (beta + sum_j gamma^j * col_{i,j} + gamma^w * table_id),  B = mult / (beta + sum_j gamma^j * table_j)
 * lookup_vars: [reps*w][n]; table_id: [n]; tables: [w+1][n]; out_A: [reps][2][n]; out_B: [2][n].
 * table_id == NULL is LookupParameters::UseSpecializedColumnsWithTableIdAsVariable (lookup_argument_in_ext.rs:354-366): every
 * sub-argument owns w + 1 variable columns, the last of them carrying the table id — lookup_vars is [reps*(w+1)][n] and no
 * constant column takes part (table_id_column_idxes is empty, setup.rs:970-971). */
void orc_lookup_polys(const uint64_t *lookup_vars, const uint64_t *table_id, const uint64_t *tables, const uint64_t *mult,
                      size_t reps, size_t w, unsigned log_n, const uint64_t *beta2, const uint64_t *gamma2,
                      uint64_t *out_A, uint64_t *out_B, int threads) {
    size_t n = (size_t)1 << log_n;
    gl2_t beta = gl2_make(gl_canon(beta2[0]), gl_canon(beta2[1])), gamma = gl2_make(gl_canon(gamma2[0]), gl_canon(gamma2[1]));
    gl2_t gp[16];
    gp[0] = gl2_make(1, 0);
    for (size_t j = 1; j <= w; j++) gp[j] = gl2_mul(gp[j - 1], gamma);
#pragma omp parallel for schedule(static) num_threads(threads)
    for (size_t r = 0; r < n; r++) {
        for (size_t i = 0; i < reps; i++) {
            gl2_t acc = beta;
            const size_t cps = table_id ? w : w + 1;   /* variable columns per sub-argument */
            for (size_t j = 0; j < cps; j++) acc = gl2_add(acc, gl2_mul_base(gp[j], gl_canon(lookup_vars[(i * cps + j) * n + r])));
            if (table_id) acc = gl2_add(acc, gl2_mul_base(gp[w], gl_canon(table_id[r])));
            gl2_t a = gl2_inv(acc);
            out_A[(2 * i) * n + r] = a.c0; out_A[(2 * i + 1) * n + r] = a.c1;
        }
        gl2_t acc = beta;
        for (size_t j = 0; j <= w; j++) acc = gl2_add(acc, gl2_mul_base(gp[j], gl_canon(tables[j * n + r])));
        gl2_t b = gl2_mul_base(gl2_inv(acc), gl_canon(mult[r]));
        out_B[r] = b.c0; out_B[n + r] = b.c1;
    }
}

/* Quotient numerator / vanishing over the first q cosets of the LDE.  Every *_lde array is [cols][Q], Q = q*n, flat
 * index I = coset*n + i with i bit-reversed.  With coset_count != 0 the arrays hold only the cosets [coset_begin,
 * coset_begin + coset_count) of those q (stride coset_count*n) and out_q is [2][coset_count*n]: the quotient is pointwise and
 * z(omega x) stays inside a coset, so a caller short of memory evaluates it coset by coset (oracle/prover_streaming.py).  alphas: [n_alpha][2] in the reference's order
 * (lookup terms | specialized gate terms (none) | general purpose gate terms | (z-1)*L1 | copy-permutation chain).
 * gates: per gate 12 ints {kind, path_len, reps, var_stride, const_stride, num_terms, path[0..5]}. */
typedef struct { int kind, path_len, reps, var_stride, const_stride, num_terms, path[6]; } orc_gate;

void orc_quotient(const uint64_t *vars, size_t V, const uint64_t *consts, size_t Kc, const uint64_t *sigmas,
                  const uint64_t *z, const uint64_t *partials, size_t n_partials, const uint64_t *lookA,
                  const uint64_t *lookB, const uint64_t *mult, const uint64_t *tables, size_t lookup_reps,
                  size_t lookup_w, size_t lookup_var_offset, size_t table_id_col, const int *gates_flat, size_t n_gates,
                  const uint64_t *non_res, size_t chunk, unsigned log_n, unsigned log_q, unsigned coset_begin,
                  const uint64_t *alphas, size_t n_alphas, const uint64_t *beta2, const uint64_t *gamma2,
                  const uint64_t *lbeta2, const uint64_t *lgamma2, uint64_t *out_q, int threads, size_t coset_count) {
    size_t n = (size_t)1 << log_n, q = (size_t)1 << log_q;
    (void)Kc;                                  /* number of constant columns: part of the call shape, the gates index into them */
    if (coset_count == 0) { coset_begin = 0; coset_count = q; }
    const size_t Q = coset_count * n;          /* stride of the arrays and number of points evaluated here */
    const orc_gate *gates = (const orc_gate *)gates_flat;
    gl2_t beta = gl2_make(beta2[0], beta2[1]), gamma = gl2_make(gamma2[0], gamma2[1]);
    gl2_t lbeta = gl2_make(lbeta2[0], lbeta2[1]), lgamma = gl2_make(lgamma2[0], lgamma2[1]);
    gl2_t lgp[16];
    lgp[0] = gl2_make(1, 0);
    for (size_t j = 1; j <= lookup_w; j++) lgp[j] = gl2_mul(lgp[j - 1], lgamma);
    size_t n_lookup_terms = lookup_reps ? lookup_reps + 1 : 0;
    size_t n_gate_terms = 0;
    for (size_t g = 0; g < n_gates; g++) n_gate_terms += (size_t)gates[g].reps * gates[g].num_terms;
    size_t n_chunks = (V + chunk - 1) / chunk;
    /* alphas consumed: lookup | gates | 1 (L1) | n_chunks (copy-perm) */
    if (n_alphas != n_lookup_terms + n_gate_terms + 1 + n_chunks) abort();
    const uint64_t *a_lookup = alphas, *a_gates = alphas + 2 * n_lookup_terms,
                   *a_l1 = a_gates + 2 * n_gate_terms, *a_cp = a_l1 + 2;
    unsigned log_Q = log_n + log_q;
    gl_t wQ = gl_omega(log_Q);
#pragma omp parallel for schedule(static) num_threads(threads)
    for (size_t I = 0; I < Q; I++) {
        size_t coset = I >> log_n, i_br = I & (n - 1);   /* coset: local to the arrays */
        gl_t x = gl_mul(GL_GEN, gl_pow(wQ, bitrev64((size_t)coset_begin * n + I, log_Q)));
        gl2_t acc = gl2_make(0, 0);
        /* ---- gates over general purpose columns: sum_g sel_g * sum_t alpha_t * term_t ---- */
        size_t aoff = 0;
        for (size_t g = 0; g < n_gates; g++) {
            const orc_gate *G = &gates[g];
            if (G->num_terms == 0) continue;
            if (G->kind >= 5) {   /* op-list / Poseidon2 flattened gates: added by oracle/prover.py from the program's own semantics */
                aoff += (size_t)G->reps * G->num_terms;
                continue;
            }
            gl_t sel = 1;
            for (int b = 0; b < G->path_len; b++) {
                gl_t c = consts[(size_t)b * Q + I];
                sel = gl_mul(sel, G->path[b] ? c : gl_sub(1, c));
            }
            gl2_t gsum = gl2_make(0, 0);
            for (int r = 0; r < G->reps; r++) {
                size_t vb = (size_t)r * G->var_stride, cb = (size_t)G->path_len + (size_t)r * G->const_stride;
#define VAR(k) vars[(vb + (k)) * Q + I]
#define CON(k) consts[(cb + (k)) * Q + I]
                gl_t term;
                if (G->kind == 1) {            /* ConstantsAllocator: a - c */
                    term = gl_sub(VAR(0), CON(0));
                } else if (G->kind == 2) {     /* FMA without constant: q*a*b + l*c - d  (constants shared by the row) */
                    gl_t qc = consts[((size_t)G->path_len) * Q + I], lc = consts[((size_t)G->path_len + 1) * Q + I];
                    term = gl_sub(gl_add(gl_mul(VAR(2), lc), gl_mul(qc, gl_mul(VAR(0), VAR(1)))), VAR(3));
                } else {                       /* Reduction<4>: sum c_i v_i - r */
                    term = 0;
                    for (int k = 0; k < 4; k++) term = gl_add(term, gl_mul(VAR(k), consts[((size_t)G->path_len + k) * Q + I]));
                    term = gl_sub(term, VAR(4));
                }
#undef VAR
#undef CON
                gl2_t al = gl2_make(a_gates[2 * aoff], a_gates[2 * aoff + 1]);
                gsum = gl2_add(gsum, gl2_mul_base(al, term));
                aoff++;
            }
            acc = gl2_add(acc, gl2_mul_base(gsum, sel));
        }
        /* ---- (z(x) - 1) * L1~(x),  L1~ = (x^n - 1)/(x - 1) ---- */
        gl2_t zv = gl2_make(z[I], z[Q + I]);
        {
            gl_t l1 = gl_mul(gl_sub(gl_pow(x, n), 1), gl_inv(gl_sub(x, 1)));
            gl2_t t = gl2_mul_base(gl2_make(gl_sub(zv.c0, 1), zv.c1), l1);
            acc = gl2_add(acc, gl2_mul(t, gl2_make(a_l1[0], a_l1[1])));
        }
        /* ---- copy permutation chain: lhs = [partials..., z(omega x)], rhs = [z, partials...] ---- */
        {
            size_t i_nat = bitrev64(i_br, log_n);
            size_t i_next = bitrev64((i_nat + 1) & (n - 1), log_n);
            size_t In = coset * n + i_next;
            gl2_t z_shift = gl2_make(z[In], z[Q + In]);
            for (size_t j = 0; j < n_chunks; j++) {
                gl2_t lhs = (j + 1 < n_chunks) ? gl2_make(partials[(2 * j) * Q + I], partials[(2 * j + 1) * Q + I]) : z_shift;
                gl2_t rhs = (j == 0) ? zv : gl2_make(partials[(2 * (j - 1)) * Q + I], partials[(2 * (j - 1) + 1) * Q + I]);
                for (size_t c = j * chunk; c < (j + 1) * chunk && c < V; c++) {
                    gl_t w = vars[c * Q + I], s = sigmas[c * Q + I];
                    gl2_t d = gl2_make(gl_add(gl_add(gl_mul(s, beta.c0), w), gamma.c0), gl_add(gl_mul(s, beta.c1), gamma.c1));
                    lhs = gl2_mul(lhs, d);
                    gl_t kx = gl_mul(x, non_res[c]);
                    gl2_t nm = gl2_make(gl_add(gl_add(gl_mul(kx, beta.c0), w), gamma.c0), gl_add(gl_mul(kx, beta.c1), gamma.c1));
                    rhs = gl2_mul(rhs, nm);
                }
                gl2_t t = gl2_mul(gl2_sub(lhs, rhs), gl2_make(a_cp[2 * j], a_cp[2 * j + 1]));
                acc = gl2_add(acc, t);
            }
            (void)n_partials;
        }
        /* ---- lookups: A_i*(sum gamma^j col_j + gamma^w*tid + beta) - 1 ;  B*(sum gamma^j table_j + beta) - mult ---- */
        if (lookup_reps) {
            for (size_t i = 0; i < lookup_reps; i++) {
                gl2_t d = lbeta;
                /* table_id_col == (size_t)-1: the table id is the (w+1)-th variable column of the sub-argument
                 * (UseSpecializedColumnsWithTableIdAsVariable, lookup_argument_in_ext.rs:949-1000: capacity = w + 1, no constant) */
                const int tid_var = table_id_col == (size_t)-1;
                const size_t cps = tid_var ? lookup_w + 1 : lookup_w;
                for (size_t j = 0; j < cps; j++)
                    d = gl2_add(d, gl2_mul_base(lgp[j], vars[(lookup_var_offset + i * cps + j) * Q + I]));
                if (!tid_var) d = gl2_add(d, gl2_mul_base(lgp[lookup_w], consts[table_id_col * Q + I]));
                gl2_t t = gl2_mul(gl2_make(lookA[(2 * i) * Q + I], lookA[(2 * i + 1) * Q + I]), d);
                t.c0 = gl_sub(t.c0, 1);
                acc = gl2_add(acc, gl2_mul(t, gl2_make(a_lookup[2 * i], a_lookup[2 * i + 1])));
            }
            gl2_t d = lbeta;
            for (size_t j = 0; j <= lookup_w; j++) d = gl2_add(d, gl2_mul_base(lgp[j], tables[j * Q + I]));
            gl2_t t = gl2_mul(gl2_make(lookB[I], lookB[Q + I]), d);
            t.c0 = gl_sub(t.c0, mult[I]);
            acc = gl2_add(acc, gl2_mul(t, gl2_make(a_lookup[2 * lookup_reps], a_lookup[2 * lookup_reps + 1])));
        }
        /* ---- divide by the vanishing polynomial x^n - 1 (constant on a coset) ---- */
        gl_t vinv = gl_inv(gl_sub(gl_pow(x, n), 1));
        acc = gl2_mul_base(acc, vinv);
        out_q[I] = acc.c0; out_q[Q + I] = acc.c1;
    }
}
