// C-ABI entry points of seam S2 for the prover's second round and its quotient terms (rows a9, a10, a12 of SURVEY §8 as
// stand-alone operators): the kernels bj_prove runs, callable on caller-provided device columns so that each can be
// compared with the reference function it replaces (and with oracle/prover_ops.c in tests/test_gpu_stage_ops.py).
#include "ctx.h"

#include <vector>

using gl::u64;

namespace bj {
void launch_copy_perm_stage2(const u64 *d_vars, size_t var_stride, const u64 *d_sigmas, size_t sig_stride,
                             const u64 *d_non_res, unsigned V, unsigned chunk, unsigned log_n, const u64 *d_tw_fwd,
                             const u64 *beta, const u64 *gamma, u64 *d_tmp, u64 *d_z, u64 *d_partials, hipStream_t s, bool small_non_residues);
void launch_lookup_polys(const u64 *d_lvars, size_t var_stride, const u64 *d_table_id, const u64 *d_tables,
                         size_t tab_stride, const u64 *d_mult, unsigned reps, unsigned w, unsigned log_n,
                         const u64 *beta, const u64 *gamma, u64 *d_A, u64 *d_B, hipStream_t s);
void launch_quotient_gates(const u64 *d_vars, size_t var_stride, const u64 *d_consts, size_t const_stride,
                           const int *h_gates_flat, unsigned n_gates, const u64 *d_alphas, size_t Q, u64 *d_out0,
                           u64 *d_out1, hipStream_t s);
void launch_quotient_lookup(const u64 *d_lvars, size_t var_stride, const u64 *d_table_id, const u64 *d_tables,
                            size_t tab_stride, const u64 *d_mult, const u64 *d_A, const u64 *d_B, size_t s2_stride,
                            unsigned reps, unsigned w, const u64 *lbeta, const u64 *lgamma, const u64 *d_alphas,
                            size_t Q, u64 *d_out0, u64 *d_out1, hipStream_t s);
bool launch_combine_residues(const u64 *d_residues, unsigned W, size_t E, unsigned n_cols, const u64 *h_a, u64 *d_out, hipStream_t s);
void launch_quotient_copy_perm(const u64 *d_vars, size_t var_stride, const u64 *d_sigmas, size_t sig_stride,
                               const u64 *d_stage2, size_t s2_stride, const u64 *d_non_res, unsigned V, unsigned chunk,
                               unsigned log_n, unsigned log_L, const u64 *d_tw_fwd, const u64 *beta, const u64 *gamma,
                               const u64 *alpha_l1, const u64 *d_alphas_cp, size_t Q_local, size_t I0, const u64 *d_inv_xm1, u64 *d_out0,
                               u64 *d_out1, hipStream_t s, bool small_non_residues);
}  // namespace bj

namespace {
struct Tmp {   // short-lived device block; freed after the stream has drained (these are test / plumbing entry points)
    bj_ctx *ctx;
    void *p = nullptr;
    bool arena = false;
    explicit Tmp(bj_ctx *c) : ctx(c) {}
    int alloc(size_t bytes) {
        p = bj::tmp_alloc(ctx, bytes, &arena);
        return p ? BJ_OK : bj::fail(ctx, BJ_ERR_OOM, "out of device memory (%zu bytes)", bytes);
    }
    ~Tmp() {
        if (p) {
            (void)hipStreamSynchronize(ctx->stream);
            bj::tmp_free(ctx, p, arena);
        }
    }
};
}  // namespace

extern "C" {

int bj_copy_perm_stage2(bj_ctx *ctx, const uint64_t *d_vars, size_t var_stride, const uint64_t *d_sigmas, size_t sig_stride,
                        const uint64_t *h_non_residues, unsigned num_vars, unsigned chunk, unsigned log_n,
                        const uint64_t *h_beta, const uint64_t *h_gamma, uint64_t *d_z, uint64_t *d_partials) {
    if (int rc = bj::bind(ctx)) return rc;
    if (!d_vars || !d_sigmas || !h_non_residues || !h_beta || !h_gamma || !d_z)
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_copy_perm_stage2: null pointer");
    if (num_vars == 0 || chunk == 0 || log_n > 30) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_copy_perm_stage2: bad geometry");
    const size_t n = (size_t)1 << log_n;
    if (var_stride < n || sig_stride < n) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_copy_perm_stage2: column stride below n");
    const unsigned n_chunks = (num_vars + chunk - 1) / chunk;
    if (n_chunks > 1 && !d_partials) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_copy_perm_stage2: null partial-product buffer");
    if (int rc = bj::ensure_twiddles(ctx, log_n, false)) return rc;
    Tmp nr(ctx), tmp(ctx), dummy(ctx);
    if (int rc = nr.alloc(8 * (size_t)num_vars)) return rc;
    if (int rc = tmp.alloc(8 * ((size_t)2 * n_chunks * n + 2 * ((n + 1023) / 1024) + 16))) return rc;
    u64 *partials = d_partials;
    if (!partials) {   // one chunk: the kernel still wants a valid pointer
        if (int rc = dummy.alloc(64)) return rc;
        partials = (u64 *)dummy.p;
    }
    if (int rc = bj::h2d_async(ctx, nr.p, h_non_residues, 8 * (size_t)num_vars)) return rc;
    bool small_k = true;
    for (unsigned c = 0; c < num_vars; c++) small_k = small_k && gl::canon(h_non_residues[c]) < ((u64)1 << 32);
    bj::launch_copy_perm_stage2(d_vars, var_stride, d_sigmas, sig_stride, (const u64 *)nr.p, num_vars, chunk, log_n, ctx->tw_fwd,
                                h_beta, h_gamma, (u64 *)tmp.p, d_z, partials, ctx->stream, small_k);
    BJ_CHECK_LAUNCH(ctx);
    return BJ_OK;
}

int bj_lookup_polys(bj_ctx *ctx, const uint64_t *d_lookup_vars, size_t var_stride, const uint64_t *d_table_id,
                    const uint64_t *d_tables, size_t table_stride, const uint64_t *d_multiplicities, unsigned reps, unsigned width,
                    unsigned log_n, const uint64_t *h_beta, const uint64_t *h_gamma, uint64_t *d_A, uint64_t *d_B) {
    if (int rc = bj::bind(ctx)) return rc;
    // d_table_id may be NULL: the table id is then the (width+1)-th variable column of every sub-argument
    if (!d_lookup_vars || !d_tables || !d_multiplicities || !h_beta || !h_gamma || !d_A || !d_B)
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_lookup_polys: null pointer");
    if (reps == 0 || width == 0 || width > 8 || log_n > 30) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_lookup_polys: bad geometry (width 1..8)");
    const size_t n = (size_t)1 << log_n;
    if (var_stride < n || table_stride < n) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_lookup_polys: column stride below n");
    bj::launch_lookup_polys(d_lookup_vars, var_stride, d_table_id, d_tables, table_stride, d_multiplicities, reps, width, log_n,
                            h_beta, h_gamma, d_A, d_B, ctx->stream);
    BJ_CHECK_LAUNCH(ctx);
    return BJ_OK;
}

int bj_quotient_gates(bj_ctx *ctx, const uint64_t *d_vars, size_t var_stride, unsigned num_gp_vars, const uint64_t *d_consts,
                      size_t const_stride, unsigned num_constant_cols, const bj_gate_desc *gates, unsigned num_gates,
                      const uint64_t *h_alphas, size_t num_points, uint64_t *d_out0, uint64_t *d_out1) {
    if (int rc = bj::bind(ctx)) return rc;
    if (!d_vars || !d_consts || !gates || !d_out0 || !d_out1) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_quotient_gates: null pointer");
    if (num_gates == 0 || num_gates > 16) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_quotient_gates: 1..16 gate types");
    if (var_stride < num_points || const_stride < num_points)
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_quotient_gates: column stride below the number of points");
    std::vector<int> flat;
    size_t n_terms = 0;
    for (unsigned g = 0; g < num_gates; g++) {
        const bj_gate_desc &G = gates[g];
        if (G.kind < BJ_GATE_CONSTANT_ALLOCATOR || G.kind > BJ_GATE_NOP)
            return bj::fail(ctx, BJ_ERR_UNSUPPORTED, "bj_quotient_gates: gate %u: only the hand-written evaluators (kinds 1..4); op lists go "
                                                     "through bj_gate_program_eval", g);
        if (G.path_len > 6 || (G.kind != BJ_GATE_NOP && G.num_terms != 1))
            return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_quotient_gates: bad gate descriptor %u", g);
        static const unsigned width[5] = {0, 1, 4, 5, 0};
        const size_t last_rep = G.num_repetitions ? G.num_repetitions - 1 : 0;
        // ConstantsAllocator reads one constant per repetition, the other two evaluators row-shared constants after the path
        const size_t const_end = G.kind == BJ_GATE_CONSTANT_ALLOCATOR ? G.path_len + last_rep * G.const_stride + 1
                                 : G.kind == BJ_GATE_FMA_NO_CONSTANT  ? G.path_len + 2
                                 : G.kind == BJ_GATE_REDUCTION4       ? G.path_len + 4 : G.path_len;
        if (G.kind != BJ_GATE_NOP && G.num_repetitions &&
            (last_rep * G.var_stride + width[G.kind] > num_gp_vars || const_end > num_constant_cols))
            return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_quotient_gates: gate %u reads past the given columns", g);
        if (G.path_len > num_constant_cols)
            return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_quotient_gates: gate %u: selector path longer than the constant columns", g);
        int f[12] = {G.kind, (int)G.path_len, (int)G.num_repetitions, (int)G.var_stride, (int)G.const_stride, (int)G.num_terms,
                     0, 0, 0, 0, 0, 0};
        for (unsigned b = 0; b < G.path_len; b++) f[6 + b] = G.path[b] ? 1 : 0;
        flat.insert(flat.end(), f, f + 12);
        n_terms += (size_t)G.num_repetitions * G.num_terms;
    }
    if (n_terms && !h_alphas) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_quotient_gates: null alpha powers");
    Tmp al(ctx);
    if (int rc = al.alloc(16 * (n_terms + 1))) return rc;
    if (n_terms)
        if (int rc = bj::h2d_async(ctx, al.p, h_alphas, 16 * n_terms)) return rc;
    bj::launch_quotient_gates(d_vars, var_stride, d_consts, const_stride, flat.data(), num_gates, (const u64 *)al.p, num_points,
                              d_out0, d_out1, ctx->stream);
    BJ_CHECK_LAUNCH(ctx);
    return BJ_OK;
}

int bj_quotient_lookup(bj_ctx *ctx, const uint64_t *d_lookup_vars, size_t var_stride, const uint64_t *d_table_id,
                       const uint64_t *d_tables, size_t table_stride, const uint64_t *d_multiplicities, const uint64_t *d_A,
                       const uint64_t *d_B, size_t stage2_stride, unsigned reps, unsigned width, const uint64_t *h_beta,
                       const uint64_t *h_gamma, const uint64_t *h_alphas, size_t num_points, uint64_t *d_out0, uint64_t *d_out1) {
    if (int rc = bj::bind(ctx)) return rc;
    if (!d_lookup_vars || !d_tables || !d_multiplicities || !d_A || !d_B || !h_beta || !h_gamma || !h_alphas ||
        !d_out0 || !d_out1)   // d_table_id may be NULL (table id as the last variable column of a sub-argument)
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_quotient_lookup: null pointer");
    if (reps == 0 || width == 0 || width > 8) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_quotient_lookup: bad geometry (width 1..8)");
    if (var_stride < num_points || table_stride < num_points || stage2_stride < num_points)
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_quotient_lookup: column stride below the number of points");
    Tmp al(ctx);
    if (int rc = al.alloc(16 * ((size_t)reps + 1))) return rc;
    if (int rc = bj::h2d_async(ctx, al.p, h_alphas, 16 * ((size_t)reps + 1))) return rc;
    bj::launch_quotient_lookup(d_lookup_vars, var_stride, d_table_id, d_tables, table_stride, d_multiplicities, d_A, d_B,
                               stage2_stride, reps, width, h_beta, h_gamma, (const u64 *)al.p, num_points, d_out0, d_out1,
                               ctx->stream);
    BJ_CHECK_LAUNCH(ctx);
    return BJ_OK;
}

int bj_quotient_copy_perm(bj_ctx *ctx, const uint64_t *d_vars, size_t var_stride, const uint64_t *d_sigmas, size_t sig_stride,
                          const uint64_t *d_stage2, size_t stage2_stride, const uint64_t *h_non_residues, unsigned num_vars,
                          unsigned chunk, unsigned log_n, unsigned log_lde, const uint64_t *h_beta, const uint64_t *h_gamma,
                          const uint64_t *h_alphas, size_t num_points, size_t first_point, uint64_t *d_out0, uint64_t *d_out1) {
    if (int rc = bj::bind(ctx)) return rc;
    if (!d_vars || !d_sigmas || !d_stage2 || !h_non_residues || !h_beta || !h_gamma || !h_alphas || !d_out0 || !d_out1)
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_quotient_copy_perm: null pointer");
    if (num_vars == 0 || num_vars > 4096 || chunk == 0 || log_n > 30 || log_lde > 6)
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_quotient_copy_perm: bad geometry (1..4096 columns, LDE factor at most 64)");
    const size_t n = (size_t)1 << log_n;
    if (first_point + num_points > (n << log_lde)) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_quotient_copy_perm: points past the LDE domain");
    if (var_stride < num_points || sig_stride < num_points || stage2_stride < num_points)
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_quotient_copy_perm: column stride below the number of points");
    if (int rc = bj::ensure_twiddles(ctx, log_n + log_lde, false)) return rc;
    const unsigned n_chunks = (num_vars + chunk - 1) / chunk;
    Tmp nr(ctx), al(ctx);
    if (int rc = nr.alloc(8 * (size_t)num_vars)) return rc;
    if (int rc = al.alloc(16 * ((size_t)n_chunks + 1))) return rc;
    if (int rc = bj::h2d_async(ctx, nr.p, h_non_residues, 8 * (size_t)num_vars)) return rc;
    if (int rc = bj::h2d_async(ctx, al.p, h_alphas + 2, 16 * (size_t)n_chunks)) return rc;
    bool small_k = true;
    for (unsigned c = 0; c < num_vars; c++) small_k = small_k && gl::canon(h_non_residues[c]) < ((u64)1 << 32);
    bj::launch_quotient_copy_perm(d_vars, var_stride, d_sigmas, sig_stride, d_stage2, stage2_stride, (const u64 *)nr.p, num_vars,
                                  chunk, log_n, log_lde, ctx->tw_fwd, h_beta, h_gamma, h_alphas, (const u64 *)al.p, num_points,
                                  first_point, nullptr, d_out0, d_out1, ctx->stream, small_k);
    BJ_CHECK_LAUNCH(ctx);
    return BJ_OK;
}

int bj_combine_residues(bj_ctx *ctx, const uint64_t *d_residues, unsigned world, size_t residue_len, unsigned num_cols,
                        const uint64_t *h_moduli, uint64_t *d_out) {
    if (int rc = bj::bind(ctx)) return rc;
    if (!d_residues || !h_moduli || !d_out) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_combine_residues: null pointer");
    if (world == 0 || world > 8 || residue_len == 0 || num_cols == 0 || num_cols > 65535)
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_combine_residues: 1..8 residues of at least one coefficient, at most 65535 columns");
    if (!bj::launch_combine_residues(d_residues, world, residue_len, num_cols, h_moduli, d_out, ctx->stream))
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_combine_residues: the moduli x^E - a_i are not pairwise distinct");
    BJ_CHECK_LAUNCH(ctx);
    return BJ_OK;
}

}  // extern "C"
