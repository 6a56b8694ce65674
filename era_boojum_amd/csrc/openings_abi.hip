// C-ABI entry points for the opening stage (barycentric evaluation at z, DEEP quotient).  Kernels: openings.hip.
#include "ctx.h"

#include <vector>

using gl::u64;

namespace bj {
void launch_barycentric_weights(u64 *d_w0, u64 *d_w1, const u64 *d_tw_fwd, unsigned log_n, u64 coset, const u64 *at,
                                hipStream_t s);
unsigned barycentric_num_blocks(size_t n);
void launch_barycentric_eval(const u64 *const *d_col_ptrs, unsigned n_cols, size_t n, const u64 *d_w0, const u64 *d_w1,
                             u64 *d_partials, u64 *d_out, hipStream_t s);
void launch_deep_accumulate(const u64 *const *d_col_ptrs, const u64 *d_coefs, unsigned n_cols, size_t N, size_t I0,
                            const u64 *d_tw_fwd, u64 c0, u64 c1, u64 at0, u64 at1, u64 *d_dst0, u64 *d_dst1,
                            int accumulate, hipStream_t s);
}  // namespace bj

namespace bj {
void launch_linear_combination(const u64 *const *d_col_ptrs, const u64 *d_coefs, unsigned n_cols, size_t n, u64 *d_out0,
                               u64 *d_out1, hipStream_t s);
int combine_monomials(bj_ctx *ctx, const uint64_t *const *h_src_c0, const uint64_t *const *h_src_c1, size_t n_src,
                      const uint64_t *h_challenges, size_t n, uint64_t *d_out0, uint64_t *d_out1);
int deep_accumulate_range(bj_ctx *ctx, const uint64_t *const *h_src_c0, const uint64_t *const *h_src_c1, size_t n_src,
                          const uint64_t *h_values, const uint64_t *h_challenges, const uint64_t *at2, unsigned log_n,
                          unsigned log_lde, size_t N_local, size_t I0, uint64_t *d_dst_c0, uint64_t *d_dst_c1,
                          int accumulate);
struct DeepSetHost {   // one opening set, host side: sources (device pointers), values and challenges as F_p^2 pairs, the point
    const uint64_t *const *src_c0, *const *src_c1;
    size_t n_src;
    const uint64_t *values, *challenges, *at2;
};
int deep_accumulate_multi(bj_ctx *ctx, const DeepSetHost *sets, unsigned n_sets, unsigned log_n, unsigned log_lde, size_t N_local,
                          size_t I0, uint64_t *d_dst_c0, uint64_t *d_dst_c1, int accumulate);
}
namespace {
// device-side argument block: [ptrs (n_cols)] [coefs (2*n_cols)] in one temporary allocation
struct DevArgs {
    bj_ctx *ctx = nullptr;
    void *d = nullptr;
    bool from_arena = false;
    int alloc(bj_ctx *c, size_t bytes) {
        ctx = c;
        d = bj::tmp_alloc(c, bytes, &from_arena);
        return d ? BJ_OK : bj::fail(c, BJ_ERR_OOM, "argument block allocation failed");
    }
    ~DevArgs() {
        if (d) bj::tmp_free(ctx, d, from_arena);
    }
};
}  // namespace

extern "C" {

int bj_barycentric_weights(bj_ctx *ctx, unsigned log_n, uint64_t coset, const uint64_t *at2, uint64_t *d_w0,
                           uint64_t *d_w1) {
    if (int rc = bj::bind(ctx)) return rc;
    if (!at2 || !d_w0 || !d_w1) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_barycentric_weights: null pointer");
    if (log_n > 30) return bj::fail(ctx, BJ_ERR_UNSUPPORTED, "bj_barycentric_weights: log_n > 30");
    if (gl::canon(coset) == 0) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_barycentric_weights: zero coset");
    if (int rc = bj::ensure_twiddles(ctx, log_n ? log_n : 1, false)) return rc;
    bj::launch_barycentric_weights(d_w0, d_w1, ctx->tw_fwd, log_n, coset, at2, ctx->stream);
    BJ_CHECK_LAUNCH(ctx);
    return BJ_OK;
}

int bj_barycentric_eval_batch(bj_ctx *ctx, const uint64_t *const *h_col_ptrs, unsigned n_cols, unsigned log_n,
                              const uint64_t *d_w0, const uint64_t *d_w1, uint64_t *h_out) {
    if (int rc = bj::bind(ctx)) return rc;
    if (n_cols == 0) return BJ_OK;
    if (!h_col_ptrs || !d_w0 || !d_w1 || !h_out)
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_barycentric_eval_batch: null pointer");
    if (log_n > 30) return bj::fail(ctx, BJ_ERR_UNSUPPORTED, "bj_barycentric_eval_batch: log_n > 30");
    for (unsigned c = 0; c < n_cols; c++)
        if (!h_col_ptrs[c]) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_barycentric_eval_batch: null column %u", c);
    const size_t n = (size_t)1 << log_n;
    const unsigned nb = bj::barycentric_num_blocks(n);
    DevArgs args;
    size_t bytes = (size_t)n_cols * sizeof(u64 *) + ((size_t)n_cols * nb * 2 + (size_t)n_cols * 2) * sizeof(u64);
    if (int rc = args.alloc(ctx, bytes)) return rc;
    const u64 **d_ptrs = (const u64 **)args.d;
    u64 *d_partials = (u64 *)(d_ptrs + n_cols);
    u64 *d_out = d_partials + (size_t)n_cols * nb * 2;
    if (int rc = bj::h2d_async(ctx, (void *)d_ptrs, h_col_ptrs, n_cols * sizeof(u64 *))) return rc;
    // barycentric, SURVEY §8d: 8 n per base column + 16 n of weights (cached across the columns of a workgroup)
    const int pb = bj::probe_begin(ctx, "barycentric_eval", 8.0 * (double)n * n_cols + 16.0 * (double)n);
    bj::launch_barycentric_eval(d_ptrs, n_cols, n, d_w0, d_w1, d_partials, d_out, ctx->stream);
    bj::probe_end(ctx, pb);
    BJ_CHECK_LAUNCH(ctx);
    return bj_memcpy_d2h(ctx, h_out, d_out, (size_t)n_cols * 2 * sizeof(u64));
}

int bj_deep_quotient_accumulate(bj_ctx *ctx, const uint64_t *const *h_src_c0, const uint64_t *const *h_src_c1,
                                size_t n_src, const uint64_t *h_values, const uint64_t *h_challenges,
                                const uint64_t *at2, unsigned log_n, unsigned log_lde, uint64_t *d_dst_c0,
                                uint64_t *d_dst_c1, int accumulate) {
    return bj::deep_accumulate_range(ctx, h_src_c0, h_src_c1, n_src, h_values, h_challenges, at2, log_n, log_lde,
                                     (size_t)1 << (log_n + log_lde), 0, d_dst_c0, d_dst_c1, accumulate);
}

}  // extern "C"

namespace bj {
// out = sum_k ch_k * src_k over n entries (sources: base columns, or F_p^2 columns as (c0, c1) pairs) — same flattening
// of F_p^2 sources into base columns with F_p^2 coefficients as the DEEP call below
int combine_monomials(bj_ctx *ctx, const uint64_t *const *h_src_c0, const uint64_t *const *h_src_c1, size_t n_src,
                      const uint64_t *h_challenges, size_t n, uint64_t *d_out0, uint64_t *d_out1) {
    if (int rc = bj::bind(ctx)) return rc;
    std::vector<const u64 *> ptrs;
    std::vector<u64> coefs;
    for (size_t k = 0; k < n_src; k++) {
        gl::e2 ch{gl::canon(h_challenges[2 * k]), gl::canon(h_challenges[2 * k + 1])};
        ptrs.push_back(h_src_c0[k]);
        coefs.push_back(ch.c0);
        coefs.push_back(ch.c1);
        if (h_src_c1 && h_src_c1[k]) {
            ptrs.push_back(h_src_c1[k]);
            coefs.push_back(gl::mul(gl::GEN, ch.c1));
            coefs.push_back(ch.c0);
        }
    }
    const unsigned n_cols = (unsigned)ptrs.size();
    DevArgs args;
    if (int rc = args.alloc(ctx, n_cols * sizeof(u64 *) + coefs.size() * sizeof(u64))) return rc;
    const u64 **d_ptrs = (const u64 **)args.d;
    u64 *d_coefs = (u64 *)(d_ptrs + n_cols);
    if (int rc = bj::h2d_async(ctx, (void *)d_ptrs, ptrs.data(), n_cols * sizeof(u64 *))) return rc;
    if (int rc = bj::h2d_async(ctx, d_coefs, coefs.data(), coefs.size() * sizeof(u64))) return rc;
    bj::launch_linear_combination(d_ptrs, d_coefs, n_cols, n, d_out0, d_out1, ctx->stream);
    BJ_CHECK_LAUNCH(ctx);
    if (!args.from_arena) BJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BJ_OK;
}

// the same over the LOCAL index range [I0, I0 + N_local) of the LDE domain (a contiguous range of cosets owned by one GPU);
// source and destination pointers address the local range
int deep_accumulate_range(bj_ctx *ctx, const uint64_t *const *h_src_c0, const uint64_t *const *h_src_c1, size_t n_src,
                          const uint64_t *h_values, const uint64_t *h_challenges, const uint64_t *at2, unsigned log_n,
                          unsigned log_lde, size_t N_local, size_t I0, uint64_t *d_dst_c0, uint64_t *d_dst_c1,
                          int accumulate) {
    if (int rc = bj::bind(ctx)) return rc;
    if (!h_src_c0 || !h_values || !h_challenges || !at2 || !d_dst_c0 || !d_dst_c1)
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_deep_quotient_accumulate: null pointer");
    if (n_src == 0 || n_src > (1u << 20)) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_deep_quotient_accumulate: bad source count");
    const unsigned log_full = log_n + log_lde;
    if (log_full == 0 || log_full > 32) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_deep_quotient_accumulate: bad domain");
    if (int rc = bj::ensure_twiddles(ctx, log_full, false)) return rc;
    // flatten F_p^2 sources into base columns with F_p^2 coefficients; C = sum_k ch_k * v_k
    std::vector<const u64 *> ptrs;
    std::vector<u64> coefs;
    gl::e2 C{0, 0};
    for (size_t k = 0; k < n_src; k++) {
        if (!h_src_c0[k]) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_deep_quotient_accumulate: null source %zu", k);
        gl::e2 ch{gl::canon(h_challenges[2 * k]), gl::canon(h_challenges[2 * k + 1])};
        gl::e2 v{gl::canon(h_values[2 * k]), gl::canon(h_values[2 * k + 1])};
        C = gl::e2_add(C, gl::e2_mul(ch, v));
        ptrs.push_back(h_src_c0[k]);
        coefs.push_back(ch.c0);
        coefs.push_back(ch.c1);
        if (h_src_c1 && h_src_c1[k]) {  // (f0 + f1 u) * ch = ... + f1 * (7 ch1 + ch0 u)
            ptrs.push_back(h_src_c1[k]);
            coefs.push_back(gl::mul(gl::GEN, ch.c1));
            coefs.push_back(ch.c0);
        }
    }
    const unsigned n_cols = (unsigned)ptrs.size();
    DevArgs args;
    if (int rc = args.alloc(ctx, n_cols * sizeof(u64 *) + coefs.size() * sizeof(u64))) return rc;
    const u64 **d_ptrs = (const u64 **)args.d;
    u64 *d_coefs = (u64 *)(d_ptrs + n_cols);
    if (int rc = bj::h2d_async(ctx, (void *)d_ptrs, ptrs.data(), n_cols * sizeof(u64 *))) return rc;
    if (int rc = bj::h2d_async(ctx, d_coefs, coefs.data(), coefs.size() * sizeof(u64))) return rc;
    bj::launch_deep_accumulate(d_ptrs, d_coefs, n_cols, N_local, I0, ctx->tw_fwd, C.c0, C.c1,
                               gl::canon(at2[0]), gl::canon(at2[1]), d_dst_c0, d_dst_c1, accumulate, ctx->stream);
    BJ_CHECK_LAUNCH(ctx);
    if (!args.from_arena) BJ_HIP(ctx, hipStreamSynchronize(ctx->stream));  // a hipMalloc'ed argument block is freed on return
    return BJ_OK;
}

// up to DEEP_MAX_SETS opening sets at once (same argument meaning per set as deep_accumulate_range): one launch, one inversion
// per lane for all of them, the destination written once
int deep_accumulate_multi(bj_ctx *ctx, const DeepSetHost *sets, unsigned n_sets, unsigned log_n, unsigned log_lde, size_t N_local,
                          size_t I0, uint64_t *d_dst_c0, uint64_t *d_dst_c1, int accumulate) {
    if (int rc = bj::bind(ctx)) return rc;
    if (!sets || n_sets == 0 || n_sets > (unsigned)DEEP_MAX_SETS || !d_dst_c0 || !d_dst_c1)
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "deep_accumulate_multi: bad arguments");
    const unsigned log_full = log_n + log_lde;
    if (log_full == 0 || log_full > 32) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "deep_accumulate_multi: bad domain");
    if (int rc = bj::ensure_twiddles(ctx, log_full, false)) return rc;
    std::vector<const u64 *> ptrs;
    std::vector<u64> coefs;
    DeepSetHostArgs a[DEEP_MAX_SETS];
    size_t first_col[DEEP_MAX_SETS];
    for (unsigned t = 0; t < n_sets; t++) {
        const DeepSetHost &S = sets[t];
        if (!S.src_c0 || !S.values || !S.challenges || !S.at2 || S.n_src == 0 || S.n_src > (1u << 20))
            return bj::fail(ctx, BJ_ERR_INVALID_ARG, "deep_accumulate_multi: bad set %u", t);
        first_col[t] = ptrs.size();
        gl::e2 C{0, 0};
        for (size_t k = 0; k < S.n_src; k++) {
            if (!S.src_c0[k]) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "deep_accumulate_multi: null source %zu of set %u", k, t);
            const gl::e2 ch{gl::canon(S.challenges[2 * k]), gl::canon(S.challenges[2 * k + 1])};
            const gl::e2 v{gl::canon(S.values[2 * k]), gl::canon(S.values[2 * k + 1])};
            C = gl::e2_add(C, gl::e2_mul(ch, v));
            ptrs.push_back(S.src_c0[k]);
            coefs.push_back(ch.c0);
            coefs.push_back(ch.c1);
            if (S.src_c1 && S.src_c1[k]) {
                ptrs.push_back(S.src_c1[k]);
                coefs.push_back(gl::mul(gl::GEN, ch.c1));
                coefs.push_back(ch.c0);
            }
        }
        a[t].n_cols = (unsigned)(ptrs.size() - first_col[t]);
        a[t].c0 = C.c0;
        a[t].c1 = C.c1;
        a[t].at0 = gl::canon(S.at2[0]);
        a[t].at1 = gl::canon(S.at2[1]);
    }
    DevArgs args;
    if (int rc = args.alloc(ctx, ptrs.size() * sizeof(u64 *) + coefs.size() * sizeof(u64))) return rc;
    const u64 **d_ptrs = (const u64 **)args.d;
    u64 *d_coefs = (u64 *)(d_ptrs + ptrs.size());
    if (int rc = bj::h2d_async(ctx, (void *)d_ptrs, ptrs.data(), ptrs.size() * sizeof(u64 *))) return rc;
    if (int rc = bj::h2d_async(ctx, d_coefs, coefs.data(), coefs.size() * sizeof(u64))) return rc;
    for (unsigned t = 0; t < n_sets; t++) {
        a[t].d_cols = d_ptrs + first_col[t];
        a[t].d_coefs = d_coefs + 2 * first_col[t];
    }
    bj::launch_deep_accumulate_multi(a, n_sets, N_local, I0, ctx->tw_fwd, d_dst_c0, d_dst_c1, accumulate, ctx->stream);
    BJ_CHECK_LAUNCH(ctx);
    if (!args.from_arena) BJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BJ_OK;
}
}  // namespace bj
