// Host orchestration of the FRI commit phase (do_fri) + transcript objects behind the C ABI.
//   do_fri                         src/cs/implementations/fri/mod.rs:49-358
//   compute_fri_schedule           src/cs/implementations/prover.rs:2281-2372
//   query openings                 src/cs/implementations/proof.rs:65-100, fri/mod.rs:829-895 (QuerySource)
// All polynomial data stay in HBM; the host only sees caps (cap_size*32 B), the final monomials and the challenges.
#include "ctx.h"
#include "host_transcript.hpp"
#include "fri_types.h"

#include <cstring>
#include <vector>

using gl::u64;

extern "C" {

// ---------------------------------------------------------------------------------------------- transcript ABI
int bj_transcript_create(int kind, bj_transcript **out) {
    if (!out) return BJ_ERR_INVALID_ARG;
    *out = nullptr;
    if (kind < BJ_TRANSCRIPT_POSEIDON2 || kind > BJ_TRANSCRIPT_KECCAK256) return BJ_ERR_UNSUPPORTED;
    *out = new bj_transcript();
    (*out)->t.kind = kind;
    return BJ_OK;
}
void bj_transcript_destroy(bj_transcript *t) { delete t; }
int bj_transcript_absorb(bj_transcript *t, const uint64_t *els, size_t n) {
    if (!t || (!els && n)) return BJ_ERR_INVALID_ARG;
    t->t.absorb(els, n);
    return BJ_OK;
}
int bj_transcript_absorb_cap(bj_transcript *t, const uint64_t *digest_words, size_t n_words) {
    if (!t || (!digest_words && n_words)) return BJ_ERR_INVALID_ARG;
    t->t.absorb_cap(digest_words, n_words);
    return BJ_OK;
}
int bj_transcript_challenge(bj_transcript *t, uint64_t *out) {
    if (!t || !out) return BJ_ERR_INVALID_ARG;
    *out = t->t.challenge();
    return BJ_OK;
}
int bj_transcript_query_index(bj_transcript *t, unsigned log_n, unsigned log_lde, uint64_t *out_index) {
    if (!t || !out_index || log_n + log_lde == 0 || log_n + log_lde > 40) return BJ_ERR_INVALID_ARG;
    if (t->bools.max_needed == 0) t->bools.max_needed = log_n + log_lde;
    if (t->bools.max_needed != log_n + log_lde) return BJ_ERR_INVALID_ARG;
    *out_index = t->bools.query_index(t->t, log_n, log_lde);
    return BJ_OK;
}

// ------------------------------------------------------------------------------------------------ schedule
int bj_fri_schedule(uint32_t security_bits, size_t cap_size, uint32_t pow_bits, uint32_t rate_log2,
                    uint32_t initial_degree_log2, uint32_t *new_pow_bits, size_t *num_queries, uint32_t *schedule,
                    size_t *schedule_len, size_t *final_degree) {
    if (!schedule || !schedule_len || rate_log2 == 0 || security_bits <= pow_bits || !bj::is_pow2(cap_size))
        return BJ_ERR_INVALID_ARG;
    uint32_t raw = security_bits - pow_bits, new_pow = pow_bits;
    if (raw % rate_log2 != 0 && new_pow >= rate_log2 - (raw % rate_log2)) new_pow -= rate_log2 - (raw % rate_log2);
    raw = security_bits - new_pow;
    uint32_t nq = raw / rate_log2 + (raw % rate_log2 != 0 ? 1 : 0);
    size_t stop = cap_size >> rate_log2;
    if (stop < 1) stop = 1;
    uint32_t stop_log = bj::log2_exact(stop), cap_log = bj::log2_exact(cap_size);
    uint32_t deg = initial_degree_log2;
    size_t len = 0;
    while (deg > stop_log) {
        if (deg + rate_log2 <= cap_log) break;
        uint32_t gap = deg - stop_log;
        uint32_t k = gap >= 3 ? 3 : gap;
        deg -= k;
        schedule[len++] = k;
        if (k == 1 && gap == 1) break;
        if (deg + rate_log2 <= cap_log) break;
    }
    *schedule_len = len;
    if (new_pow_bits) *new_pow_bits = new_pow;
    if (num_queries) *num_queries = nq;
    if (final_degree) *final_degree = (size_t)1 << deg;
    return BJ_OK;
}

// -------------------------------------------------------------------------------------------------- do_fri
void bj_fri_destroy(bj_fri *f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    for (void *p : f->owned) (void)hipFree(p);
    delete f;
}

int bj_fri_prove(bj_ctx *ctx, const uint64_t *d_c0, const uint64_t *d_c1, unsigned log_n, unsigned log_lde,
                 const uint32_t *schedule, size_t schedule_len, size_t cap_size, bj_transcript *tr, bj_fri **out) {
    return bj::fri_prove_sharded(ctx, bj::Shard{}, d_c0, d_c1, log_n, log_lde, schedule, schedule_len, cap_size, tr, out);
}

}  // extern "C"

// With sh.world > 1 the codeword arrives split by contiguous index ranges (d_c0/d_c1 address this rank's N/world values).
// Oracle 0 is committed shard-wise (local subtree, gathered cap) and folded locally — a fold of 2^k adjacent values
// never crosses a shard boundary — then the folded layer (N/2^k0 values) is all-gathered and the remaining, geometrically
// smaller oracles are computed by every rank, so all ranks hold the same transcript.
int bj::fri_prove_sharded(bj_ctx *ctx, const bj::Shard &sh, const u64 *d_c0, const u64 *d_c1, unsigned log_n,
                          unsigned log_lde, const uint32_t *schedule, size_t schedule_len, size_t cap_size,
                          bj_transcript *tr, bj_fri **out) {
    if (int rc = bj::bind(ctx)) return rc;
    if (!out) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_fri_prove: null out pointer");
    *out = nullptr;
    if (!d_c0 || !d_c1 || !schedule || !tr || schedule_len == 0 || schedule_len > 31)
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_fri_prove: null/empty argument");
    if (log_lde == 0 || log_n + log_lde > 32 || !bj::is_pow2(cap_size))
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_fri_prove: bad domain / cap size");
    const unsigned log_full = log_n + log_lde;
    unsigned total_fold = 0;
    for (size_t i = 0; i < schedule_len; i++) {
        if (schedule[i] < 1 || schedule[i] > 3)
            return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_fri_prove: folding steps must be 1..3 (fri/mod.rs:204-205)");
        total_fold += schedule[i];
    }
    if (total_fold > log_n) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_fri_prove: schedule folds below degree 1");
    if (sh.world > 1 && (!bj::is_pow2(sh.world) || cap_size % sh.world || ((size_t)1 << log_full) >> schedule[0] < cap_size))
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_fri_prove: world must be a power of two dividing the cap size");
    if (int rc = bj::ensure_twiddles(ctx, log_full, true)) return rc;  // roots of the FULL domain (fri/mod.rs:192)

    bj_fri *f = new bj_fri();
    f->device = ctx->device;
    f->cap_size = cap_size;
    f->log_full = log_full;
    f->log_lde = log_lde;
    const u64 *cur0 = d_c0, *cur1 = d_c1;
    size_t cur_len = (size_t)1 << log_full;
    u64 kappa = gl::inv(gl::GEN);  // coset_inverse, squared after every fold
    int rc = BJ_OK;
    auto bail = [&](int code) {
        bj_fri_destroy(f);
        return code;
    };
    // layers and trees live as long as the bj_fri object: arena blocks inside a proof (released with the arena), hipMalloc
    // blocks (listed in f->owned) for the stand-alone bj_fri_prove
    auto alloc = [&](size_t elems) -> u64 * {
        bool from_arena = false;
        u64 *p = (u64 *)bj::tmp_alloc(ctx, elems * sizeof(u64), &from_arena);
        if (p && !from_arena) f->owned.push_back(p);
        return p;
    };
    for (size_t step = 0; step < schedule_len; step++) {
        const unsigned k = schedule[step];
        const unsigned parts = step == 0 ? sh.world : 1;   // how many ranks share this oracle
        const size_t loc_len = cur_len / parts, loc_cap = cap_size / parts;
        bj_fri::Oracle o;
        o.d_c0 = (u64 *)cur0;
        o.d_c1 = (u64 *)cur1;
        o.len = loc_len;
        o.log_e = k;
        o.num_leaves = loc_len >> k;
        o.world = parts;
        if (o.num_leaves < loc_cap || loc_cap == 0)
            return bail(bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_fri_prove: oracle smaller than cap"));
        size_t nd = 2 * o.num_leaves - loc_cap;
        if (!(o.d_tree = alloc(nd * 4))) return bail(bj::fail(ctx, BJ_ERR_OOM, "bj_fri_prove: tree allocation failed"));
        f->oracles.push_back(o);
        bj_fri::Oracle &oo = f->oracles.back();
        // oracle: 2^k values of c0 then of c1 per leaf (merkle_tree.rs:176-386)
        rc = bj_merkle_tree_build_chunked(ctx, cur0, cur1, loc_len, k, loc_cap, oo.d_tree);
        if (rc) return bail(rc);
        oo.cap.resize(4 * cap_size);
        if (parts == 1)
            rc = bj_merkle_tree_cap(ctx, oo.d_tree, oo.num_leaves, cap_size, oo.cap.data());
        else
            rc = bj::gather_cap(ctx, sh, oo.d_tree, oo.num_leaves, cap_size, oo.cap.data());
        if (rc) return bail(rc);
        tr->t.absorb_cap(oo.cap.data(), oo.cap.size());
        oo.ch0 = tr->t.challenge();
        oo.ch1 = tr->t.challenge();
        // fold by 2^k in one fused launch; alpha and kappa are squared per inner fold inside the kernel
        const size_t out_len = cur_len >> k, loc_out = loc_len >> k;
        u64 *nxt = alloc(2 * out_len);
        if (!nxt) return bail(bj::fail(ctx, BJ_ERR_OOM, "bj_fri_prove: layer allocation failed"));
        // FRI fold by 2^k over m F_p^2 inputs, SURVEY §8d: 16 m + 16 m / 2^k
        const int pf = bj::probe_begin(ctx, "fri_fold_first", 16.0 * (double)loc_len * (1.0 + 1.0 / (double)(1u << k)));
        if (parts == 1) {
            bj::launch_fri_fold_step(cur0, cur1, cur_len, k, nxt, nxt + out_len, ctx->tw_inv, kappa, oo.ch0, oo.ch1,
                                     ctx->stream);
            bj::probe_end(ctx, pf);
        } else {
            u64 *part = alloc(2 * loc_out);   // [2][loc_out] of this rank, gathered into nxt = [2][out_len]
            if (!part) return bail(bj::fail(ctx, BJ_ERR_OOM, "bj_fri_prove: layer allocation failed"));
            bj::launch_fri_fold_step(cur0, cur1, loc_len, k, part, part + loc_out, ctx->tw_inv, kappa, oo.ch0, oo.ch1,
                                     ctx->stream, (size_t)sh.rank * loc_out);
            bj::probe_end(ctx, pf);
            rc = bj::all_gather_columns(ctx, sh, part, nxt, 2, loc_out);
            if (rc) return bail(rc);
        }
        if (hipGetLastError() != hipSuccess) return bail(bj::fail(ctx, BJ_ERR_HIP, "bj_fri_prove: fold launch failed"));
        for (unsigned i = 0; i < k; i++) kappa = gl::sqr(kappa);
        cur0 = nxt;
        cur1 = nxt + out_len;
        cur_len = out_len;
        if (step + 1 == schedule_len) {
            f->d_last0 = nxt;
            f->d_last1 = nxt + out_len;
            f->last_len = out_len;
        }
    }
    // final interpolation: bit-reverse, iNTT on coset kappa^-1, keep len/lde coefficients (fri/mod.rs:312-343)
    const unsigned log_m = bj::log2_exact(cur_len);
    u64 *fin = alloc(2 * cur_len);
    if (!fin) return bail(bj::fail(ctx, BJ_ERR_OOM, "bj_fri_prove: final buffer allocation failed"));
    rc = bj_bitreverse_batch(ctx, cur0, fin, log_m, 2, cur_len);
    if (!rc) rc = bj_intt_batch(ctx, fin, fin, log_m, 2, cur_len, gl::inv(kappa));
    f->final_c0.resize(cur_len);
    f->final_c1.resize(cur_len);
    if (!rc) rc = bj_memcpy_d2h(ctx, f->final_c0.data(), fin, cur_len * sizeof(u64));
    if (!rc) rc = bj_memcpy_d2h(ctx, f->final_c1.data(), fin + cur_len, cur_len * sizeof(u64));
    if (rc) return bail(rc);
    f->final_degree = cur_len >> log_lde;
    // the reference asserts the high coefficients vanish (fri/mod.rs:327-336): report instead of panicking
    for (size_t i = f->final_degree; i < cur_len; i++)
        if (f->final_c0[i] || f->final_c1[i])
            return bail(bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_fri_prove: codeword is not low degree (final monomials do not vanish)"));
    tr->t.absorb(f->final_c0.data(), f->final_degree);
    tr->t.absorb(f->final_c1.data(), f->final_degree);
    // oracles after the first own their source as one allocation starting at d_c0
    *out = f;
    return BJ_OK;
}

extern "C" {

size_t bj_fri_num_oracles(const bj_fri *f) { return f ? f->oracles.size() : 0; }
size_t bj_fri_final_degree(const bj_fri *f) { return f ? f->final_degree : 0; }

int bj_fri_cap(const bj_fri *f, size_t oracle, uint64_t *h_cap) {
    if (!f || oracle >= f->oracles.size() || !h_cap) return BJ_ERR_INVALID_ARG;
    std::memcpy(h_cap, f->oracles[oracle].cap.data(), f->oracles[oracle].cap.size() * sizeof(u64));
    return BJ_OK;
}
int bj_fri_challenge(const bj_fri *f, size_t oracle, uint64_t *h_ch2) {
    if (!f || oracle >= f->oracles.size() || !h_ch2) return BJ_ERR_INVALID_ARG;
    h_ch2[0] = f->oracles[oracle].ch0;
    h_ch2[1] = f->oracles[oracle].ch1;
    return BJ_OK;
}
int bj_fri_final_monomials(const bj_fri *f, uint64_t *h_c0, uint64_t *h_c1) {
    if (!f || !h_c0 || !h_c1) return BJ_ERR_INVALID_ARG;
    std::memcpy(h_c0, f->final_c0.data(), f->final_degree * sizeof(u64));
    std::memcpy(h_c1, f->final_c1.data(), f->final_degree * sizeof(u64));
    return BJ_OK;
}
// Opening of oracle `oracle` at FLAT index `index` of that oracle's source array (index = coset*domain + inner of
// that layer; the caller shifts the global query index right by the folds applied so far).  Returns the leaf's
// 2*E elements (c0 values then c1 values) and its Merkle path.
int bj_fri_query(bj_ctx *ctx, const bj_fri *f, size_t oracle, size_t index, uint64_t *h_leaf_elements,
                 uint64_t *h_path) {
    if (int rc = bj::bind(ctx)) return rc;
    if (!f || oracle >= f->oracles.size() || !h_leaf_elements)
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_fri_query: bad oracle index / null pointer");
    const bj_fri::Oracle &o = f->oracles[oracle];
    if (o.world > 1) return bj::fail(ctx, BJ_ERR_UNSUPPORTED, "bj_fri_query: oracle is sharded across GPUs (use bj_prove)");
    if (index >= o.len) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_fri_query: index out of range");
    const size_t E = (size_t)1 << o.log_e, leaf = index >> o.log_e;
    BJ_HIP(ctx, hipMemcpyAsync(h_leaf_elements, o.d_c0 + leaf * E, E * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    BJ_HIP(ctx, hipMemcpyAsync(h_leaf_elements + E, o.d_c1 + leaf * E, E * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    BJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < 2 * E; i++) h_leaf_elements[i] = gl::canon(h_leaf_elements[i]);
    u64 digest[4];
    return bj_merkle_tree_proof(ctx, o.d_tree, o.num_leaves, f->cap_size, leaf, digest, h_path);
}

}  // extern "C"
