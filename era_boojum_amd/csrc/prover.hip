// Whole-prover orchestration: bj_setup_create[_sharded] / bj_prove_dev / bj_prove (seam S1 of SURVEY.md §8b), on one GPU
// or with the LDE cosets of one proof split across several (bj_comm, include/boojum_hip.h).
// Follows prove_cpu_basic (src/cs/implementations/prover.rs:153-2266) round by round; the host only runs the
// Fiat–Shamir transcript and O(#columns) scalar arithmetic, every polynomial stays in HBM.
//
// Circuit class: general-purpose gates ConstantsAllocator / FMA-without-constant / Reduction<4> / Nop selected by a
// selector tree over the first constant columns, specialized lookups with a shared constant table id
// (LookupParameters::UseSpecializedColumnsWithTableIdAsConstant) or no lookups, no witness columns, Poseidon2 tree
// hasher + Poseidon2 transcript, PoW off — the configuration of the reference's SHA-256 bench
// (src/gadgets/sha256/mod.rs:284-375).
#include "ctx.h"
#include "host_transcript.hpp"
#include "fri_types.h"
#include "gate_program.h"

#include <chrono>
#include <cstring>
#include <deque>
#include <vector>

using gl::u64;

namespace bj {
// stage2.hip
void launch_copy_perm_stage2(const u64 *d_vars, size_t var_stride, const u64 *d_sigmas, size_t sig_stride,
                             const u64 *d_non_res, unsigned V, unsigned chunk, unsigned log_n, const u64 *d_tw_fwd,
                             const u64 *beta, const u64 *gamma, u64 *d_tmp, u64 *d_z, u64 *d_partials, hipStream_t s, bool small_non_residues);
void launch_lookup_polys(const u64 *d_lvars, size_t var_stride, const u64 *d_table_id, const u64 *d_tables,
                         size_t tab_stride, const u64 *d_mult, unsigned reps, unsigned w, unsigned log_n,
                         const u64 *beta, const u64 *gamma, u64 *d_A, u64 *d_B, hipStream_t s);
// quotient.hip
void launch_quotient_gates(const u64 *d_vars, size_t var_stride, const u64 *d_consts, size_t const_stride,
                           const int *h_gates_flat, unsigned n_gates, const u64 *d_alphas, size_t Q, u64 *d_out0,
                           u64 *d_out1, hipStream_t s);
void launch_quotient_lookup(const u64 *d_lvars, size_t var_stride, const u64 *d_table_id, const u64 *d_tables,
                            size_t tab_stride, const u64 *d_mult, const u64 *d_A, const u64 *d_B, size_t s2_stride,
                            unsigned reps, unsigned w, const u64 *lbeta, const u64 *lgamma, const u64 *d_alphas,
                            size_t Q, u64 *d_out0, u64 *d_out1, hipStream_t s);
void launch_quotient_copy_perm(const u64 *d_vars, size_t var_stride, const u64 *d_sigmas, size_t sig_stride,
                               const u64 *d_stage2, size_t s2_stride, const u64 *d_non_res, unsigned V, unsigned chunk,
                               unsigned log_n, unsigned log_L, const u64 *d_tw_fwd, const u64 *beta, const u64 *gamma,
                               const u64 *alpha_l1, const u64 *d_alphas_cp, size_t Q_local, size_t I0, const u64 *d_inv_xm1, u64 *d_out0,
                               u64 *d_out1, hipStream_t s, bool small_non_residues);
void launch_inv_x_minus_one(const u64 *d_tw_fwd, size_t Q, size_t I0, u64 *d_out, hipStream_t s);
bool launch_combine_residues(const u64 *d_residues, unsigned W, size_t E, unsigned n_cols, const u64 *h_a, u64 *d_out, hipStream_t s);
void launch_gather_rows(const u64 *d_base, size_t col_stride, unsigned n_cols, const u64 *d_idx, unsigned n_idx,
                        u64 *d_out, hipStream_t s);
void launch_merkle_paths(const u64 *d_tree, size_t num_leaves, unsigned depth, const u64 *d_idx, unsigned n_idx,
                         u64 *d_out, hipStream_t s);
void launch_gather_fri_leaves(const u64 *d_c0, const u64 *d_c1, unsigned log_e, const u64 *d_leaf_idx, unsigned n_idx,
                              u64 *d_out, hipStream_t s);
void launch_quotient_poseidon2_flattened(const u64 *d_vars, size_t var_stride, const u64 *d_consts, size_t const_stride,
                                         unsigned path_len, const unsigned char *path, const u64 *d_alphas, size_t Q,
                                         u64 *d_out0, u64 *d_out1, hipStream_t s);
int combine_monomials(bj_ctx *ctx, const uint64_t *const *h_src_c0, const uint64_t *const *h_src_c1, size_t n_src,
                      const uint64_t *h_challenges, size_t n, uint64_t *d_out0, uint64_t *d_out1);
int deep_accumulate_range(bj_ctx *ctx, const uint64_t *const *h_src_c0, const uint64_t *const *h_src_c1, size_t n_src,
                          const uint64_t *h_values, const uint64_t *h_challenges, const uint64_t *at2, unsigned log_n,
                          unsigned log_lde, size_t N_local, size_t I0, uint64_t *d_dst_c0, uint64_t *d_dst_c1,
                          int accumulate);
struct DeepSetHost {
    const uint64_t *const *src_c0, *const *src_c1;
    size_t n_src;
    const uint64_t *values, *challenges, *at2;
};
int deep_accumulate_multi(bj_ctx *ctx, const DeepSetHost *sets, unsigned n_sets, unsigned log_n, unsigned log_lde, size_t N_local,
                          size_t I0, uint64_t *d_dst_c0, uint64_t *d_dst_c1, int accumulate);
}  // namespace bj

struct bj_setup {
    int device = 0;
    // circuit
    unsigned log_n = 0, V = 0, num_gp_vars = 0, nC = 0, lookup_w = 0, lookup_reps = 0, table_id_col = 0, q = 0;
    unsigned Wc = 0;               // non-copiable witness columns (behind the V variable columns in the witness oracle)
    bool tid_var = false;          // UseSpecializedColumnsWithTableIdAsVariable: the table id is the last of lookup_cps = lookup_w + 1
    unsigned lookup_cps = 0;       // variable columns of a sub-argument (specialized_columns_per_subargument, cs/mod.rs:300-312)
    std::vector<unsigned> gate_wit_stride;   // per general-purpose gate: per_chunk_offset.witnesses_offset
    std::vector<int> gates_flat;   // 12 ints per gate
    std::vector<bj::DevProgram> programs;   // per gate; empty (block == nullptr) unless kind == BJ_GATE_PROGRAM
    unsigned n_gates = 0;
    struct SpecGate {                        // a gate over specialized columns: op list, no selector, own columns
        bj::DevProgram program;
        unsigned reps = 0, width = 0, terms = 0, first_col = 0;
        unsigned first_const = 0, const_width = 0;   // its constant columns: reps * const_width of them from first_const on
    };
    std::vector<SpecGate> spec;
    unsigned n_spec_terms = 0;
    std::vector<u64> non_residues;
    bool small_non_residues = false;   // every k_c < 2^32 (canonical): quotient_copy_perm multiplies by them as 32-bit integers
    std::vector<unsigned> pub_cols, pub_rows;
    // proof config
    unsigned fri_lde = 0, cap_size = 0, security = 0, pow_bits = 0, transcript = BJ_TRANSCRIPT_POSEIDON2, hasher = BJ_HASHER_POSEIDON2;
    unsigned pow_runner = BJ_POW_BLAKE2S256;   // the POW type parameter of prove_cpu_basic (pow.rs:6-31)
    unsigned L = 0, log_L = 0, log_fri = 0, log_q = 0;
    unsigned n_cols = 0;           // V sigmas + nC constants + (w+1) tables
    // shard of the LDE domain held by this GPU: cosets [c0, c0 + cl), i.e. flat indices [c0*n, (c0+cl)*n)
    bj::Shard sh;
    unsigned c0 = 0, cl = 0;
    size_t Ls = 0;                 // column stride of every LDE array = cl * n
    size_t Nl = 0;                 // Merkle leaves held here (n * fri_lde / world)
    size_t cap_l = 0;              // cap nodes of the local subtree (cap_size / world)
    u64 *d_nat = nullptr;          // [n_cols][n] natural-order values (replicated)
    u64 *d_mono = nullptr;         // [n_cols][n] monomial forms (replicated; the DEEP numerator is combined on them)
    bool tiled = false;            // monomials (these and every proof's) in the tiled layout of ntt_r16.hip: 2^22-row traces
    u64 *d_lde = nullptr;          // [n_cols][cl][n]
    u64 *d_tree = nullptr;         // local subtree
    u64 *d_non_res = nullptr;
    u64 *d_inv_xm1 = nullptr;      // 1 / (x - 1) on the points this GPU evaluates the quotient on (a property of the domain)
    std::vector<u64> cap;
};

struct bj_proof {
    std::vector<u64> data;
    float stage_ms[8] = {0};
    float comm_ms = 0;             // sharded proofs: time between the start and the end of every collective on this rank, summed
    unsigned comm_calls = 0;
    size_t comm_bytes = 0;         // bytes received from the other ranks
    size_t ws_reserved = 0, ws_high_water = 0, ws_overflow_slabs = 0;   // bj_proof_workspace_bytes
    struct KernelStat {            // bj_proof_kernel_stats: first launch of each probed kernel
        const char *name;
        float ms;
        double bytes;
    } kernel_stats[BJ_MAX_KERNEL_PROBES] = {};
    unsigned n_kernel_stats = 0;
};

namespace {

struct DevBuf {
    u64 *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    int alloc(bj_ctx *ctx, size_t elems) {
        if (hipMalloc((void **)&p, (elems ? elems : 1) * sizeof(u64)) != hipSuccess)
            return bj::fail(ctx, BJ_ERR_OOM, "device allocation of %zu MiB failed", elems * 8 >> 20);
        return BJ_OK;
    }
};

// workspace buffer carved out of the context's arena (no hipMalloc/hipFree inside a proof)
struct ArenaBuf {
    u64 *p = nullptr;
    int alloc(bj_ctx *ctx, size_t elems) {
        p = bj::arena_alloc(ctx, elems ? elems : 1);
        if (!p) return bj::fail(ctx, BJ_ERR_OOM, "workspace reservation too small for %zu MiB", elems * 8 >> 20);
        return BJ_OK;
    }
};

gl::e2 e2c(const u64 *p) { return {gl::canon(p[0]), gl::canon(p[1])}; }

struct StageTimer {
    hipStream_t s;
    std::chrono::steady_clock::time_point t0;
    explicit StageTimer(hipStream_t st) : s(st) {
        (void)hipStreamSynchronize(s);
        t0 = std::chrono::steady_clock::now();
    }
    float lap() {
        (void)hipStreamSynchronize(s);
        auto t1 = std::chrono::steady_clock::now();
        float ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
        t0 = t1;
        return ms;
    }
};

}  // namespace

namespace bj {
unsigned setup_world(const bj_setup *s) { return s ? s->sh.world : 0; }
int all_gather(bj_ctx *ctx, const Shard &sh, const u64 *d_send, u64 *d_recv, size_t elems) {
    if (!elems) return BJ_OK;
    if (sh.world == 1) {
        if (d_send != d_recv)
            BJ_HIP(ctx, hipMemcpyAsync(d_recv, d_send, elems * 8, hipMemcpyDeviceToDevice, ctx->stream));
        return BJ_OK;
    }
    ctx->comm_bytes += elems * 8 * (sh.world - 1);
    if (sh.comm.all_gather_stream) {   // stream-ordered transport (in-library RCCL): no synchronisation on either side
        hipEvent_t *ev = nullptr;
        if (ctx->in_proof && ctx->comm_n < 48) {
            ev = ctx->comm_ev[ctx->comm_n];
            if (!ev[0]) (void)hipEventCreate(&ev[0]);
            if (!ev[1]) (void)hipEventCreate(&ev[1]);
            if (ev[0] && ev[1]) (void)hipEventRecord(ev[0], ctx->stream);
        }
        if (int rc = sh.comm.all_gather_stream(sh.comm.user, d_send, d_recv, elems * 8, ctx->stream))
            return fail(ctx, BJ_ERR_HIP, "sharded prover: the stream-ordered all_gather failed (%d)", rc);
        if (ev && ev[0] && ev[1]) {
            (void)hipEventRecord(ev[1], ctx->stream);
            ctx->comm_n++;
        }
        return BJ_OK;
    }
    if (!sh.comm.all_gather) return fail(ctx, BJ_ERR_INVALID_ARG, "sharded prover: no all_gather callback");
    BJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const auto t0 = std::chrono::steady_clock::now();
    if (int rc = sh.comm.all_gather(sh.comm.user, d_send, d_recv, elems * 8))
        return fail(ctx, BJ_ERR_HIP, "sharded prover: the host's all_gather callback failed (%d)", rc);
    ctx->comm_host_ms += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (ctx->in_proof) ctx->comm_n += 0x10000;   // high half: synchronous calls
    return BJ_OK;
}

int all_gather_columns(bj_ctx *ctx, const Shard &sh, const u64 *d_send, u64 *d_dst, unsigned parts, size_t part_len) {
    if (sh.world == 1) return all_gather(ctx, sh, d_send, d_dst, (size_t)parts * part_len);
    const size_t per = (size_t)parts * part_len;
    bool from_arena = false;
    u64 *tmp = (u64 *)tmp_alloc(ctx, per * sh.world * 8, &from_arena);
    if (!tmp) return fail(ctx, BJ_ERR_OOM, "all_gather_columns: staging allocation failed");
    int rc = all_gather(ctx, sh, d_send, tmp, per);
    if (!rc) {
        // tmp[r][p][part_len] -> dst[p][r][part_len]: for each rank one strided 2-D copy
        for (unsigned r = 0; r < sh.world && !rc; r++)
            if (hipMemcpy2DAsync(d_dst + (size_t)r * part_len, (size_t)sh.world * part_len * 8, tmp + (size_t)r * per,
                                 part_len * 8, part_len * 8, parts, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
                rc = fail(ctx, BJ_ERR_HIP, "all_gather_columns: device copy failed");
        (void)hipStreamSynchronize(ctx->stream);
    }
    tmp_free(ctx, tmp, from_arena);
    return rc;
}

int gather_cap(bj_ctx *ctx, const Shard &sh, const u64 *d_tree_local, size_t leaves_local, size_t cap_size, u64 *h_cap) {
    const size_t cap_l = cap_size / sh.world;
    if (sh.world == 1) return bj_merkle_tree_cap(ctx, d_tree_local, leaves_local, cap_size, h_cap);
    // the local cap fragment is the last layer of the local subtree
    const u64 *frag = d_tree_local + (2 * leaves_local - 2 * cap_l) * 4;
    u64 *d_all = ctx->d_small + 64 + 64 * 32;   // 4096 u64 reserved for this (ctx.h)
    if (cap_size * 4 > 4096) return fail(ctx, BJ_ERR_INVALID_ARG, "gather_cap: cap too large");
    if (int rc = all_gather(ctx, sh, frag, d_all, cap_l * 4)) return rc;
    return bj_memcpy_d2h(ctx, h_cap, d_all, cap_size * 32);
}
}  // namespace bj

extern "C" {

int bj_setup_set_comm(bj_setup *s, const bj_comm *comm) {
    if (!s || !comm) return BJ_ERR_INVALID_ARG;
    if (s->sh.world < 2 || comm->world != s->sh.world || comm->rank != s->sh.rank || (!comm->all_gather && !comm->all_gather_stream))
        return BJ_ERR_INVALID_ARG;   // only the transport changes: the shard this setup holds is fixed
    s->sh.comm = *comm;
    return BJ_OK;
}

void bj_setup_destroy(bj_setup *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->d_nat) (void)hipFree(s->d_nat);
    if (s->d_mono) (void)hipFree(s->d_mono);
    if (s->d_lde) (void)hipFree(s->d_lde);
    if (s->d_tree) (void)hipFree(s->d_tree);
    if (s->d_non_res) (void)hipFree(s->d_non_res);
    if (s->d_inv_xm1) (void)hipFree(s->d_inv_xm1);
    for (auto &p : s->programs) p.release();
    for (auto &g : s->spec) g.program.release();
    delete s;
}

int bj_setup_create(bj_ctx *ctx, const bj_circuit *c, const uint64_t *h_sigmas, const uint64_t *h_constants,
                    const uint64_t *h_tables, const bj_proof_config *cfg, bj_setup **out) {
    return bj_setup_create_sharded(ctx, c, h_sigmas, h_constants, h_tables, cfg, nullptr, out);
}

int bj_setup_create_sharded(bj_ctx *ctx, const bj_circuit *c, const uint64_t *h_sigmas, const uint64_t *h_constants,
                            const uint64_t *h_tables, const bj_proof_config *cfg, const bj_comm *comm, bj_setup **out) {
    if (int rc = bj::bind(ctx)) return rc;
    if (!out) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_setup_create: null out pointer");
    *out = nullptr;
    if (!c || !cfg || !h_sigmas || !h_constants) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_setup_create: null argument");
    if (c->log_n < 1 || c->log_n > 26) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_setup_create: log_n out of range");
    if (c->num_gates == 0 || c->num_gates > 16 || !c->gates) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_setup_create: 1..16 gates expected");
    if (!bj::is_pow2(c->quotient_degree) || !bj::is_pow2(cfg->fri_lde_factor) || cfg->fri_lde_factor < 2 ||
        !bj::is_pow2(cfg->cap_size))
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_setup_create: quotient degree / fri_lde_factor / cap must be powers of two");
    {
        const unsigned tk = cfg->transcript ? cfg->transcript : BJ_TRANSCRIPT_POSEIDON2, hk = cfg->tree_hasher ? cfg->tree_hasher : BJ_HASHER_POSEIDON2;
        if (tk > BJ_TRANSCRIPT_KECCAK256 || hk > BJ_HASHER_KECCAK256) return bj::fail(ctx, BJ_ERR_UNSUPPORTED, "bj_setup_create: unknown transcript / tree hasher");
        const bool byte_hasher = hk != BJ_HASHER_POSEIDON2, byte_transcript = tk == BJ_TRANSCRIPT_BLAKE2S || tk == BJ_TRANSCRIPT_KECCAK256;
        if (byte_hasher != byte_transcript)   // Transcript::CompatibleCap = TreeHasher::Output (prover.rs:153-168)
            return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_setup_create: a byte tree hasher (Blake2s / Keccak256) goes with a byte transcript and "
                                                     "the Poseidon2 tree hasher with an algebraic transcript");
    }
    if (cfg->fri_lde_factor > 64 || c->quotient_degree > 64)   // per-coset tables of the quotient kernels hold 64 entries
        return bj::fail(ctx, BJ_ERR_UNSUPPORTED, "bj_setup_create: fri_lde_factor and quotient_degree are limited to 64");
    if (c->num_public_inputs && (!c->public_input_cols || !c->public_input_rows))
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_setup_create: public input locations missing");
    for (unsigned i = 0; i < c->num_public_inputs; i++)
        if (c->public_input_cols[i] >= c->num_vars || (c->public_input_rows[i] >> c->log_n) != 0)
            return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_setup_create: public input %u at (column %u, row %u) is outside the %u x 2^%u trace",
                            i, c->public_input_cols[i], c->public_input_rows[i], c->num_vars, c->log_n);
    if (cfg->pow_bits > 32 || cfg->pow_bits >= cfg->security_level)
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_setup_create: pow_bits must be <= 32 and below the security level (pow.rs:53, prover.rs:2293)");
    if (cfg->pow_runner > BJ_POW_KECCAK256)
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_setup_create: pow_runner %u (0 / BJ_POW_BLAKE2S256 / BJ_POW_KECCAK256)", cfg->pow_runner);
    // LookupParameters::UseSpecializedColumnsWithTableIdAsVariable (cs/mod.rs:237-241): table_ids_column_idxes is empty (setup.rs:970-971)
    // and a sub-argument owns width + 1 variable columns, the last one the table id (lookup_argument_in_ext.rs:354-366, 949-1000)
    const bool tid_var = c->lookup_reps && c->table_id_col == BJ_TABLE_ID_AS_VARIABLE;
    const unsigned lookup_cps = c->lookup_width + (tid_var ? 1u : 0u);
    if (c->lookup_reps && (!h_tables || c->lookup_width == 0 || c->lookup_width > 8 || (!tid_var && c->table_id_col >= c->num_constant_cols)))
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_setup_create: bad lookup parameters");
    if ((uint64_t)c->num_vars < (uint64_t)c->num_gp_vars + (uint64_t)lookup_cps * c->lookup_reps || !c->non_residues)
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_setup_create: bad column counts");
    if (c->num_vars > 4096)   // the copy-permutation quotient keeps k_c * beta of every column in LDS (16 bytes per column)
        return bj::fail(ctx, BJ_ERR_UNSUPPORTED, "bj_setup_create: %u copiable columns, at most 4096 are supported", c->num_vars);
    unsigned n_chunks = (c->num_vars + c->quotient_degree - 1) / c->quotient_degree;
    if (n_chunks < 2) return bj::fail(ctx, BJ_ERR_UNSUPPORTED, "bj_setup_create: a single copy-permutation chunk is not supported");
    if (comm && comm->world > 1) {
        const unsigned W = comm->world;
        if (!bj::is_pow2(W) || W > 8 || comm->rank >= W || (!comm->all_gather && !comm->all_gather_stream))
            return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_setup_create_sharded: world must be a power of two <= 8, rank < world, callback set");
        if (cfg->fri_lde_factor % W || cfg->cap_size % W || c->quotient_degree > cfg->fri_lde_factor)
            return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_setup_create_sharded: world must divide fri_lde_factor and cap_size, and "
                                                     "quotient_degree must not exceed fri_lde_factor");
        const unsigned cl = cfg->fri_lde_factor / W;
        if ((((size_t)1 << c->log_n) * cl) < cfg->cap_size / W)
            return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_setup_create_sharded: shard smaller than its cap fragment");
        const size_t Qe_rank = ((size_t)c->quotient_degree << c->log_n) / W;   // every rank evaluates q n / W points of the quotient
        if (Qe_rank < 2 || !bj::is_pow2(Qe_rank))                              // and inverse-transforms them: a power of two
            return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_setup_create_sharded: q n / world = %zu quotient points per rank (a power of two >= 2 is needed)", Qe_rank);
    }
    bj_setup *s = new bj_setup();
    s->device = ctx->device;
    struct HasherGuard {   // the tree kernels of the calls below follow the proof config, then the context's own setting returns
        bj_ctx *c;
        int saved;
        ~HasherGuard() { c->hasher = saved; }
    } hasher_guard{ctx, ctx->hasher};
    ctx->hasher = cfg->tree_hasher ? (int)cfg->tree_hasher : BJ_HASHER_POSEIDON2;
    if (comm && comm->world > 1) {
        s->sh.rank = comm->rank;
        s->sh.world = comm->world;
        s->sh.comm = *comm;
    }
    s->log_n = c->log_n; s->V = c->num_vars; s->num_gp_vars = c->num_gp_vars; s->nC = c->num_constant_cols;
    s->Wc = c->num_witness_cols;
    s->lookup_w = c->lookup_width; s->lookup_reps = c->lookup_reps; s->table_id_col = tid_var ? 0 : c->table_id_col;
    s->tid_var = tid_var; s->lookup_cps = lookup_cps;
    s->q = c->quotient_degree;
    s->n_gates = c->num_gates;
    if (c->num_gates > 16) {
        bj_setup_destroy(s);
        return bj::fail(ctx, BJ_ERR_UNSUPPORTED, "bj_setup_create: %u gate types over general-purpose columns (at most 16)", c->num_gates);
    }
    for (unsigned g = 0; g < c->num_gates; g++) {
        const bj_gate_desc &G = c->gates[g];
        const bool p2 = G.kind == BJ_GATE_POSEIDON2_FLATTENED;
        if (G.kind < 1 || G.kind > BJ_GATE_POSEIDON2_FLATTENED || G.path_len > 6 || (G.kind == BJ_GATE_PROGRAM && !G.program) ||
            (p2 && (G.num_terms != 118 || G.num_repetitions != 1 || c->num_gp_vars < 130)) ||
            (G.kind != BJ_GATE_PROGRAM && G.kind != BJ_GATE_NOP && !p2 && G.num_terms != 1)) {
            bj_setup_destroy(s);
            return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_setup_create: bad gate descriptor %u", g);
        }
        {   // every column index the evaluator will form must exist: (reps - 1) * stride + the widest operand, after the selector path
            unsigned var_extent = 0, const_extent = 0, wit_extent = 0;
            bool const_per_rep = true;
            switch (G.kind) {
                case BJ_GATE_CONSTANT_ALLOCATOR: var_extent = 1; const_extent = 1; break;
                case BJ_GATE_FMA_NO_CONSTANT: var_extent = 4; const_extent = 2; const_per_rep = false; break;
                case BJ_GATE_REDUCTION4: var_extent = 5; const_extent = 4; const_per_rep = false; break;
                case BJ_GATE_POSEIDON2_FLATTENED: var_extent = 130; break;
                case BJ_GATE_PROGRAM: bj::gate_program_extent(G.program, &var_extent, &const_extent, &wit_extent); break;
                default: break;
            }
            s->gate_wit_stride.push_back(G.wit_stride);
            if (wit_extent && (size_t)(G.num_repetitions ? G.num_repetitions - 1 : 0) * G.wit_stride + wit_extent > c->num_witness_cols) {
                bj_setup_destroy(s);
                return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_setup_create: gate %u reads witness column %zu of %u", g,
                                (size_t)(G.num_repetitions - 1) * G.wit_stride + wit_extent, c->num_witness_cols);
            }
            const size_t last = G.num_repetitions ? G.num_repetitions - 1 : 0;
            const size_t var_end = var_extent ? last * G.var_stride + var_extent : 0;
            const size_t const_end = G.path_len + (const_extent ? (const_per_rep ? last * G.const_stride : 0) + const_extent : 0);
            if (G.kind != BJ_GATE_NOP && (G.num_repetitions == 0 || var_end > c->num_gp_vars || const_end > c->num_constant_cols)) {
                bj_setup_destroy(s);
                return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_setup_create: gate %u reads variable column %zu / constant column %zu of %u / %u "
                                "(repetitions x stride + operand index, after a selector path of %u)", g, var_end, const_end,
                                c->num_gp_vars, c->num_constant_cols, G.path_len);
            }
            if (G.path_len > c->num_constant_cols) {
                bj_setup_destroy(s);
                return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_setup_create: gate %u: selector path longer than the constant columns", g);
            }
        }
        s->programs.emplace_back();
        if (G.kind == BJ_GATE_PROGRAM) {
            if (G.program->num_writes != G.num_terms) {
                bj_setup_destroy(s);
                return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_setup_create: gate %u: program writes %u terms, descriptor says %u", g,
                                G.program->num_writes, G.num_terms);
            }
            if (int prc = s->programs.back().upload(ctx, G.program)) {
                bj_setup_destroy(s);
                return prc;
            }
        }
        int f[12] = {G.kind, (int)G.path_len, (int)G.num_repetitions, (int)G.var_stride, (int)G.const_stride,
                     (int)G.num_terms, 0, 0, 0, 0, 0, 0};
        for (unsigned b = 0; b < G.path_len; b++) f[6 + b] = G.path[b] ? 1 : 0;
        s->gates_flat.insert(s->gates_flat.end(), f, f + 12);
    }
    {   // gates over specialized columns (evaluator_data.rs:124-240, prover.rs:635-800): their variable columns follow the lookup ones
        // in declaration order; their constant columns follow the general-purpose gates' ones and the table-id column (which is the
        // first "special purpose" constant: setup.rs:963-1010), num_repetitions * const_stride columns each — every repetition its
        // own principal_width.num_constants columns (share_constants = false, per_repetition_offset.constants_offset = that width)
        // (64-bit sums: the sizes are the caller's, a wrapped 32-bit total must not pass the range checks)
        uint64_t col = (uint64_t)c->num_gp_vars + (uint64_t)lookup_cps * c->lookup_reps;
        uint64_t spec_consts = 0;
        for (unsigned g = 0; c->specialized_gates && g < c->num_specialized_gates; g++) {
            const uint64_t per_gate = (uint64_t)c->specialized_gates[g].num_repetitions * c->specialized_gates[g].const_stride;
            if (per_gate > c->num_constant_cols) { spec_consts = (uint64_t)c->num_constant_cols + 1; break; }
            spec_consts += per_gate;
        }
        if (spec_consts > c->num_constant_cols ||
            (c->lookup_reps && !tid_var && (uint64_t)c->table_id_col + 1 + spec_consts != c->num_constant_cols)) {
            bj_setup_destroy(s);
            return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_setup_create: %u constant columns declared; the specialized gates' %llu must be the "
                            "last ones, right behind the table-id column", c->num_constant_cols, (unsigned long long)spec_consts);
        }
        unsigned ccol = c->num_constant_cols - (unsigned)spec_consts;
        s->spec.resize(c->num_specialized_gates);
        for (unsigned g = 0; g < c->num_specialized_gates; g++) {
            const bj_gate_desc &G = c->specialized_gates[g];
            bool ok = c->specialized_gates && G.kind == BJ_GATE_PROGRAM && G.program && G.path_len == 0 && G.num_repetitions &&
                      G.var_stride && G.program->num_writes == G.num_terms;
            unsigned ve = 0, ce = 0, we = 0;
            if (ok) {
                bj::gate_program_extent(G.program, &ve, &ce, &we);
                // a repetition reads its own var_stride variable columns and its own const_stride constant columns, no witness column.
                // Constants SHARED by the repetitions (share_constants = true with constants) are refused: the reference itself hands
                // such an evaluator an empty constant range (per_repetition_offset.constants_offset = 0, prover.rs:748-772)
                ok = ve <= G.var_stride && we == 0 && ce <= G.const_stride;
            }
            if (!ok) {
                bj_setup_destroy(s);
                return bj::fail(ctx, BJ_ERR_UNSUPPORTED, "bj_setup_create: specialized gate %u must be an op list without a selector path "
                                "whose repetitions each read their own var_stride variable and const_stride constant columns "
                                "(share_constants = false) and no witness column", g);
            }
            bj_setup::SpecGate &sg = s->spec[g];
            if (int prc = sg.program.upload(ctx, G.program)) {
                bj_setup_destroy(s);
                return prc;
            }
            if (col + (uint64_t)G.num_repetitions * G.var_stride > c->num_vars) {
                bj_setup_destroy(s);
                return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_setup_create: specialized gate %u runs past the %u declared variable columns", g,
                                c->num_vars);
            }
            sg.reps = G.num_repetitions; sg.width = G.var_stride; sg.terms = G.num_terms; sg.first_col = (unsigned)col;
            sg.first_const = ccol; sg.const_width = G.const_stride;
            col += (uint64_t)sg.reps * sg.width;
            ccol += sg.reps * sg.const_width;
            s->n_spec_terms += sg.reps * sg.terms;
        }
        if (col != c->num_vars) {
            bj_setup_destroy(s);
            return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_setup_create: %u variable columns declared, geometry + lookups + "
                            "specialized gates make %llu", c->num_vars, (unsigned long long)col);
        }
    }
    s->non_residues.assign(c->non_residues, c->non_residues + c->num_vars);
    s->small_non_residues = true;
    for (u64 k : s->non_residues) s->small_non_residues = s->small_non_residues && gl::canon(k) < ((u64)1 << 32);
    for (unsigned i = 0; i < c->num_public_inputs; i++) {
        s->pub_cols.push_back(c->public_input_cols[i]);
        s->pub_rows.push_back(c->public_input_rows[i]);
    }
    s->fri_lde = cfg->fri_lde_factor; s->cap_size = cfg->cap_size; s->security = cfg->security_level; s->pow_bits = cfg->pow_bits;
    s->pow_runner = cfg->pow_runner ? cfg->pow_runner : BJ_POW_BLAKE2S256;
    s->transcript = cfg->transcript ? cfg->transcript : BJ_TRANSCRIPT_POSEIDON2;
    s->hasher = cfg->tree_hasher ? cfg->tree_hasher : BJ_HASHER_POSEIDON2;
    s->L = s->fri_lde > s->q ? s->fri_lde : s->q;   // used_lde_degree (prover.rs:313)
    s->log_L = bj::log2_exact(s->L); s->log_fri = bj::log2_exact(s->fri_lde); s->log_q = bj::log2_exact(s->q);
    const size_t n = (size_t)1 << s->log_n;
    const unsigned nT = s->lookup_reps ? s->lookup_w + 1 : 0;
    s->n_cols = s->V + s->nC + nT;
    s->cl = s->L / s->sh.world;
    s->c0 = s->sh.rank * s->cl;
    s->Ls = (size_t)s->cl * n;
    s->Nl = n * s->fri_lde / s->sh.world;
    s->cap_l = s->cap_size / s->sh.world;
    int rc = BJ_OK;
    auto bail = [&](int code) {
        bj_setup_destroy(s);
        return code;
    };
    if (hipMalloc((void **)&s->d_nat, (size_t)s->n_cols * n * 8) != hipSuccess ||
        hipMalloc((void **)&s->d_lde, (size_t)s->n_cols * s->Ls * 8) != hipSuccess ||
        hipMalloc((void **)&s->d_non_res, s->V * 8) != hipSuccess)
        return bail(bj::fail(ctx, BJ_ERR_OOM, "bj_setup_create: device allocation failed"));
    // leaf order of the setup oracle: sigma || constants || tables (polynomial_storage.rs:667-676)
    rc = bj_memcpy_h2d(ctx, s->d_nat, h_sigmas, (size_t)s->V * n * 8);
    if (!rc) rc = bj_memcpy_h2d(ctx, s->d_nat + (size_t)s->V * n, h_constants, (size_t)s->nC * n * 8);
    if (!rc && nT) rc = bj_memcpy_h2d(ctx, s->d_nat + (size_t)(s->V + s->nC) * n, h_tables, (size_t)nT * n * 8);
    if (!rc) rc = bj_memcpy_h2d(ctx, s->d_non_res, s->non_residues.data(), s->V * 8);
    if (rc) return bail(rc);
    {   // monomials (kept), LDE into d_lde
        if (hipMalloc((void **)&s->d_mono, (size_t)s->n_cols * n * 8) != hipSuccess)
            return bail(bj::fail(ctx, BJ_ERR_OOM, "bj_setup_create: device allocation failed"));
        s->tiled = bj::mono_tiled(s->log_n);
        rc = s->tiled ? bj::intt_to_tiled(ctx, s->d_nat, n, s->d_mono, n, s->log_n, s->n_cols)
                      : bj_intt_batch(ctx, s->d_nat, s->d_mono, s->log_n, s->n_cols, n, 1);
        if (!rc) rc = bj::lde_cosets_strided(ctx, s->d_mono, n, s->d_lde, s->Ls, s->log_n, s->n_cols, s->log_L, s->c0, s->cl, s->tiled);
        if (!rc) rc = bj_sync(ctx);
        if (rc) return bail(rc);
    }
    {   // the points the quotient is evaluated on here (prove_impl: Qe, I0) and 1 / (x - 1) on them, for the L_1 term
        const size_t Qe = (n * s->q) / s->sh.world;
        if (Qe) {
            if ((rc = bj::ensure_twiddles(ctx, s->log_n + s->log_L, false))) return bail(rc);
            if (hipMalloc((void **)&s->d_inv_xm1, Qe * 8) != hipSuccess)
                return bail(bj::fail(ctx, BJ_ERR_OOM, "bj_setup_create: device allocation failed"));
            bj::launch_inv_x_minus_one(ctx->tw_fwd, Qe, (size_t)s->c0 * n, s->d_inv_xm1, ctx->stream);
            if (hipGetLastError() != hipSuccess) return bail(bj::fail(ctx, BJ_ERR_HIP, "bj_setup_create: launch failed"));
        }
    }
    if (hipMalloc((void **)&s->d_tree, bj_merkle_tree_digests(s->Nl, s->cap_l) * 32) != hipSuccess)
        return bail(bj::fail(ctx, BJ_ERR_OOM, "bj_setup_create: tree allocation failed"));
    rc = bj_merkle_tree_build(ctx, s->d_lde, s->Ls, s->n_cols, s->Nl, s->cap_l, s->d_tree);
    s->cap.resize(4 * s->cap_size);
    if (!rc) rc = bj::gather_cap(ctx, s->sh, s->d_tree, s->Nl, s->cap_size, s->cap.data());
    if (rc) return bail(rc);
    *out = s;
    return BJ_OK;
}

int bj_setup_shape(const bj_setup *s, unsigned *log_n, unsigned *num_vars, unsigned *num_witness_cols, unsigned *num_public_inputs) {
    if (s && num_public_inputs) *num_public_inputs = (unsigned)s->pub_cols.size();
    if (!s) return BJ_ERR_INVALID_ARG;
    if (log_n) *log_n = s->log_n;
    if (num_vars) *num_vars = s->V;
    if (num_witness_cols) *num_witness_cols = s->Wc;
    return BJ_OK;
}

int bj_setup_cap(const bj_setup *s, uint64_t *h_cap) {
    if (!s || !h_cap) return BJ_ERR_INVALID_ARG;
    std::memcpy(h_cap, s->cap.data(), s->cap.size() * 8);
    return BJ_OK;
}

void bj_proof_destroy(bj_proof *p) { delete p; }
size_t bj_proof_size_u64(const bj_proof *p) { return p ? p->data.size() : 0; }
int bj_proof_serialize(const bj_proof *p, uint64_t *out) {
    if (!p || !out) return BJ_ERR_INVALID_ARG;
    std::memcpy(out, p->data.data(), p->data.size() * 8);
    return BJ_OK;
}
int bj_proof_comm_stats(const bj_proof *p, float *ms_in_collectives, size_t *calls, size_t *bytes_received) {
    if (!p) return BJ_ERR_INVALID_ARG;
    if (ms_in_collectives) *ms_in_collectives = p->comm_ms;
    if (calls) *calls = p->comm_calls;
    if (bytes_received) *bytes_received = p->comm_bytes;
    return BJ_OK;
}
int bj_proof_kernel_stats(const bj_proof *p, unsigned index, const char **name, float *ms, double *algorithmic_bytes) {
    if (!p || index >= p->n_kernel_stats) return BJ_ERR_INVALID_ARG;
    if (name) *name = p->kernel_stats[index].name;
    if (ms) *ms = p->kernel_stats[index].ms;
    if (algorithmic_bytes) *algorithmic_bytes = p->kernel_stats[index].bytes;
    return BJ_OK;
}
int bj_proof_workspace_bytes(const bj_proof *p, size_t *reserved, size_t *high_water, size_t *overflow_slabs) {
    if (!p) return BJ_ERR_INVALID_ARG;
    if (reserved) *reserved = p->ws_reserved;
    if (high_water) *high_water = p->ws_high_water;
    if (overflow_slabs) *overflow_slabs = p->ws_overflow_slabs;
    return BJ_OK;
}
int bj_setup_device_bytes(const bj_setup *s, size_t *bytes) {
    if (!s || !bytes) return BJ_ERR_INVALID_ARG;
    const size_t n = (size_t)1 << s->log_n;
    size_t b = 0;
    if (s->d_nat) b += (size_t)s->n_cols * n * 8;
    if (s->d_mono) b += (size_t)s->n_cols * n * 8;
    if (s->d_lde) b += (size_t)s->n_cols * s->Ls * 8;
    if (s->d_tree) b += bj_merkle_tree_digests(s->Nl, s->cap_l) * 32;
    if (s->d_non_res) b += (size_t)s->V * 8;
    if (s->d_inv_xm1) b += (n * s->q) / s->sh.world * 8;
    *bytes = b;
    return BJ_OK;
}
int bj_proof_stage_ms(const bj_proof *p, float *out8) {
    if (!p || !out8) return BJ_ERR_INVALID_ARG;
    std::memcpy(out8, p->stage_ms, sizeof(p->stage_ms));
    return BJ_OK;
}

}  // extern "C"

namespace {
// bj_prove: the witness is still in host memory when the proof starts.  Its columns are copied in groups on a second stream,
// every group followed by an event; the witness round below waits for a group right before it transforms it, so the PCIe
// transfer of the later groups runs under the iNTT / LDE of the earlier ones (the host buffers should be pinned).
struct HostWitness {
    const uint64_t *h_variables, *h_multiplicities;
    unsigned group;        // columns per group
    bool no_absorb;        // transfer and transform in groups, hash once at the end (a lane of bj_prove_async whose sibling is proving)
};
int prove_impl(bj_ctx *ctx, const bj_setup *S, const uint64_t *d_variables, const uint64_t *d_multiplicities,
               const uint64_t *h_public_values, bj_proof **out, const HostWitness *hw);
}  // namespace

extern "C" {

int bj_prove_dev(bj_ctx *ctx, const bj_setup *S, const uint64_t *d_variables, const uint64_t *d_multiplicities,
                 const uint64_t *h_public_values, bj_proof **out) {
    return prove_impl(ctx, S, d_variables, d_multiplicities, h_public_values, out, nullptr);
}

}  // extern "C"

namespace {
int prove_impl(bj_ctx *ctx, const bj_setup *S, const uint64_t *d_variables, const uint64_t *d_multiplicities,
               const uint64_t *h_public_values, bj_proof **out, const HostWitness *hw) {
    if (int rc = bj::bind(ctx)) return rc;
    if (!out) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_prove_dev: null out pointer");
    *out = nullptr;
    if (!S || !d_variables) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_prove_dev: null argument");
    const bool has_lookup = S->lookup_reps > 0;
    if (has_lookup && !d_multiplicities) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_prove_dev: multiplicities required");
    if (!S->pub_cols.empty() && !h_public_values) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_prove_dev: public input values required");
    hipStream_t st = ctx->stream;
    const unsigned log_n = S->log_n, V = S->V, q = S->q, L = S->L, fri = S->fri_lde, cap = S->cap_size;
    const bj::Shard &sh = S->sh;
    // N / Ln: leaves / column stride HELD BY THIS GPU (the whole domain on one GPU); Q: points of the quotient domain
    const size_t n = (size_t)1 << log_n, N = S->Nl, Q = n * q, Ln = S->Ls, I0 = (size_t)S->c0 * n;
    const size_t capl = S->cap_l;
    // quotient evaluation: the q n points are split evenly, rank r evaluates on the first Qe = q n / W points of ITS OWN range of
    // the LDE (they form the coset x_{I0} * H_Qe in bit-reversed order, whether Qe is several cosets of H_n, one, or a fraction
    // of one), inverse-transforms them to T mod (x^Qe - x_{I0}^Qe), and the residues are all-gathered and combined (below)
    const bool q_local = sh.world == 1;
    const size_t Qe = Q / sh.world;   // points this rank evaluates
    const unsigned VW = V + S->Wc;                         // variables, then the non-copiable witness columns (prover.rs:317-343)
    const unsigned nW = VW + (has_lookup ? 1 : 0);
    const unsigned n_chunks = (V + q - 1) / q, n_part = n_chunks - 1;
    const unsigned nS2 = 2 * (1 + n_part) + (has_lookup ? 2 * (S->lookup_reps + 1) : 0);
    const unsigned nT = has_lookup ? S->lookup_w + 1 : 0;
    const unsigned nC = S->nC;
    int rc = BJ_OK;
    bj_proof *proof = new bj_proof();
    struct Guard {
        bj_proof *&p;
        bool ok = false;
        ~Guard() {
            if (!ok) {
                delete p;
                p = nullptr;
            }
        }
    } guard{proof};
    {   // one reservation for every buffer below (sizes mirror the allocations; +1 MiB slack per buffer for alignment)
        const size_t tree_elems = bj_merkle_tree_digests(N, capl) * 4, slack = (size_t)1 << 17;
        size_t need = (size_t)nW * Ln + (size_t)(nW + nS2) * n + tree_elems                        // wit_lde, monomials, wit_tree
                    + (size_t)nS2 * n + ((size_t)2 * n_chunks * n + 2 * ((n + 1023) / 1024) + 16)   // s2_nat, tmp
                    + (size_t)nS2 * Ln + tree_elems                                                // s2_lde, s2_tree
                    + 2 * Q + (sh.world > 1 ? 2 * Ln * (sh.world + 1) : 0) + (size_t)2 * q * N + tree_elems + 4 * n + 4 * N                       // T (+ gather staging), q_lde, q_tree, w, deep
                    + (size_t)4096 * 1024 * (sh.world > 1 ? 1 + sh.world : 1) + 64 * slack        // alphas, query gathers
                    + 4 * N + (N * sh.world) / 2                                                   // FRI layers + trees, DEEP argument blocks
                    + (sh.world > 1 ? 10 * n : 0)                                                  // sharded DEEP numerators: slices + gather staging
                    + 4 * N                                                                        // host witness hashed in groups: the leaves' capacity words — reserved whichever
                                                                                                   // entry point the proof came through: a context that alternates between bj_prove and
                                                                                                   // bj_prove_dev (the lanes of bj_prove_async do) must not re-allocate its arena (1.8 s for 64 GB)
                    + (size_t)2 * N * (1 + S->pub_cols.size())                                     // DEEP: one extended numerator per large opening set beyond the first
                    + (S->tiled ? 2 * Q : 0);                                                      // the quotient's chunks once more, in the tiled layout
        // `need` is an upper bound by construction of the list above — checked on every proof the test suite makes (the binding
        // raises when a proof had to take an overflow slab) — and a context that has seen a larger proof keeps its size
        if (need < ctx->arena_learned) need = ctx->arena_learned;
        if ((rc = bj::arena_reset(ctx, need))) return rc;
        proof->ws_reserved = ctx->arena_elems * 8;
    }
    struct InProof {   // temporaries of the ABI calls below come out of the arena while this is alive
        bj_ctx *c;
        explicit InProof(bj_ctx *x) : c(x) {
            c->in_proof = true;
            c->comm_n = 0;
            c->comm_bytes = 0;
            c->comm_host_ms = 0;
            c->probe_n = 0;
        }
        ~InProof() { c->in_proof = false; }
    } in_proof(ctx);
    struct HasherGuard {
        bj_ctx *c;
        int saved;
        ~HasherGuard() { c->hasher = saved; }
    } hasher_guard{ctx, ctx->hasher};
    struct CopyDrain {   // bj_prove: whatever way the proof ends, no queued copy may still read the caller's witness afterwards
        bj_ctx *c;
        bool active;
        ~CopyDrain() {
            if (active && c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
        }
    } copy_drain{ctx, hw != nullptr};
    ctx->hasher = (int)S->hasher;
    if (!S->pub_cols.empty()) {   // the values that go into the transcript must be the cells they claim to be (witness.rs:21-27)
        std::vector<u64> cells(S->pub_cols.size());
        for (size_t i = 0; i < cells.size(); i++) {
            const size_t at = (size_t)S->pub_cols[i] * n + S->pub_rows[i];
            if (hw)
                cells[i] = hw->h_variables[at];
            else
                BJ_HIP(ctx, hipMemcpyAsync(&cells[i], d_variables + at, 8, hipMemcpyDeviceToHost, st));
        }
        if (!hw) BJ_HIP(ctx, hipStreamSynchronize(st));
        for (size_t i = 0; i < cells.size(); i++)
            if (gl::canon(cells[i]) != gl::canon(h_public_values[i]))
                return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_prove: public input %zu: value %llu given, the witness holds %llu at (column %u, row %u)",
                                i, (unsigned long long)gl::canon(h_public_values[i]), (unsigned long long)gl::canon(cells[i]),
                                S->pub_cols[i], S->pub_rows[i]);
    }
    StageTimer timer(st);
    bj::host::Transcript tr;
    tr.kind = (int)S->transcript;
    tr.absorb_cap(S->cap.data(), S->cap.size());                           // prover.rs:211
    if (!S->pub_cols.empty()) tr.absorb(h_public_values, S->pub_cols.size());   // prover.rs:257-259
    auto challenge2 = [&](u64 *o) {
        o[0] = tr.challenge();
        o[1] = tr.challenge();
    };
    if ((rc = bj::ensure_twiddles(ctx, log_n + (L > fri ? bj::log2_exact(L) : S->log_fri), false))) return rc;

    // ---------------- round 1: witness LDE + tree (prover.rs:270-353) ----------------
    ArenaBuf wit_lde, wit_tree, mono, mono_s2;   // mono: witness monomials, mono_s2: stage-2 monomials (both kept for DEEP)
    bool hashed_in_groups = false;               // bj_prove: the witness leaves were absorbed group by group under the transfer
    if ((rc = wit_lde.alloc(ctx, (size_t)nW * Ln))) return rc;
    if ((rc = mono.alloc(ctx, (size_t)nW * n))) return rc;
    if ((rc = mono_s2.alloc(ctx, (size_t)nS2 * n))) return rc;
    // inverse transform of columns of the main domain into the monomial layout of this trace length; extension out of it
    const bool tiled = S->tiled;
    auto intt_cols = [&](const u64 *d_in, u64 *d_out, unsigned nc) -> int {
        return tiled ? bj::intt_to_tiled(ctx, d_in, n, d_out, n, log_n, nc) : bj_intt_batch(ctx, d_in, d_out, log_n, nc, n, 1);
    };
    auto lde_cols = [&](const u64 *d_m, u64 *d_out, size_t out_stride, unsigned nc, unsigned log_lde, unsigned cb, unsigned cc) -> int {
        return bj::lde_cosets_strided(ctx, d_m, n, d_out, out_stride, log_n, nc, log_lde, cb, cc, tiled);
    };
    if (!hw) {
        rc = intt_cols(d_variables, mono.p, VW);
        if (!rc && has_lookup) rc = intt_cols(d_multiplicities, mono.p + (size_t)VW * n, 1);
        if (!rc) rc = lde_cols(mono.p, wit_lde.p, Ln, nW, S->log_L, S->c0, S->cl);
    } else {
        // all copies are queued on the copy stream at once (they run back to back at PCIe speed); the proof stream picks the
        // groups up as they land.  Column nW - 1 is the multiplicity column when there are lookups.
        if (!ctx->copy_stream) BJ_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        // groups of G columns, at most 64 of them (one event each): a wide witness gets wider groups instead of an error.  With
        // the Poseidon2 tree hasher and G a multiple of the sponge's rate, a group is also absorbed into the leaf sponges as soon
        // as it is extended (the capacity words of every leaf wait in HBM between the groups), so that hashing — the dominant
        // kernel — runs under the transfer of the later groups instead of after the last one has landed.
        unsigned G = hw->group;
        if ((nW + G - 1) / G > 64) G = (nW + 63) / 64;
        const bool absorb = ctx->hasher == BJ_HASHER_POSEIDON2 && !bj::env().prove_no_absorb && !hw->no_absorb;
        if (absorb) G = (G + 7) / 8 * 8;
        // The plan: transfer / transform groups [c0, c1) with one event each, and after some of them one absorption run over the
        // columns extended since the last one.  Nothing can be hashed before the first G columns have crossed PCIe, so those go in
        // quarters (the transforms of a quarter run under the transfer of the next) and are absorbed together; from then on the
        // transfer (PCIe, ~5 ms per 8 columns of 2^22 rows) runs ahead of the hashing (~10 ms), so the groups widen to 2 G and
        // 3 G: fewer round trips of the 32-byte capacity per leaf and fewer launch tails.
        struct Grp {
            unsigned c0, c1, absorb_from;   // absorb_from == ~0u: no absorption after this group
        };
        std::vector<Grp> plan;
        const unsigned NONE = ~0u;
        for (;;) {
            plan.clear();
            if (!absorb || bj::env().prove_uniform_groups) {
                for (unsigned c0 = 0; c0 < nW; c0 += G) plan.push_back({c0, c0 + G < nW ? c0 + G : nW, absorb ? c0 : NONE});
            } else {
                const unsigned q = G / 4;   // G is a multiple of 8
                unsigned pos = 0;
                for (unsigned k = 0; k < 4 && pos < nW; k++) {
                    const unsigned c1 = pos + q < nW ? pos + q : nW;
                    plan.push_back({pos, c1, (k == 3 || c1 == nW) ? 0u : NONE});
                    pos = c1;
                }
                const unsigned widths[6] = {1, 2, 2, 3, 3, 3};
                for (unsigned k = 0; pos < nW; k++) {
                    const unsigned w = G * widths[k < 6 ? k : 5];
                    const unsigned c1 = pos + w < nW ? pos + w : nW;
                    plan.push_back({pos, c1, pos});
                    pos = c1;
                }
            }
            if (plan.size() <= 64) break;
            G = absorb ? G + 8 : G + 1;
        }
        const unsigned n_groups = (unsigned)plan.size();
        ArenaBuf capacity;
        if (absorb) {
            if ((rc = wit_tree.alloc(ctx, bj_merkle_tree_digests(N, capl) * 4))) return rc;
            if ((rc = capacity.alloc(ctx, 4 * N))) return rc;
            hashed_in_groups = true;
        }
        for (unsigned g = 0; g < n_groups; g++) {
            if (!ctx->copy_ev[g]) BJ_HIP(ctx, hipEventCreateWithFlags(&ctx->copy_ev[g], hipEventDisableTiming));
            const unsigned c0 = plan[g].c0, c1 = plan[g].c1;
            const unsigned v1 = c1 < VW ? c1 : VW;         // variable / witness columns of this group: [c0, v1)
            if (c0 < v1)
                BJ_HIP(ctx, hipMemcpyAsync(const_cast<uint64_t *>(d_variables) + (size_t)c0 * n, hw->h_variables + (size_t)c0 * n,
                                           (size_t)(v1 - c0) * n * 8, hipMemcpyHostToDevice, ctx->copy_stream));
            if (has_lookup && c1 == nW)
                BJ_HIP(ctx, hipMemcpyAsync(const_cast<uint64_t *>(d_multiplicities), hw->h_multiplicities, n * 8, hipMemcpyHostToDevice,
                                           ctx->copy_stream));
            BJ_HIP(ctx, hipEventRecord(ctx->copy_ev[g], ctx->copy_stream));
        }
        if (absorb) BJ_HIP(ctx, hipEventRecord(ctx->ev0, st));
        for (unsigned g = 0; g < n_groups && !rc; g++) {
            const unsigned c0 = plan[g].c0, c1 = plan[g].c1;
            const unsigned v1 = c1 < VW ? c1 : VW;
            BJ_HIP(ctx, hipStreamWaitEvent(st, ctx->copy_ev[g], 0));
            if (c0 < v1) rc = intt_cols(d_variables + (size_t)c0 * n, mono.p + (size_t)c0 * n, v1 - c0);
            if (!rc && has_lookup && c1 == nW) rc = intt_cols(d_multiplicities, mono.p + (size_t)VW * n, 1);
            if (!rc) rc = lde_cols(mono.p + (size_t)c0 * n, wit_lde.p + (size_t)c0 * Ln, Ln, c1 - c0, S->log_L, S->c0, S->cl);
            if (absorb && !rc && plan[g].absorb_from != NONE) {
                const unsigned a0 = plan[g].absorb_from;
                bj::launch_poseidon2_leaves_absorb(wit_lde.p + (size_t)a0 * Ln, Ln, c1 - a0, N, capacity.p, wit_tree.p, a0 == 0, c1 == nW, st);
            }
        }
        if (absorb) BJ_HIP(ctx, hipEventRecord(ctx->ev1, st));
    }
    if (rc) return rc;
    if (!hashed_in_groups && (rc = wit_tree.alloc(ctx, bj_merkle_tree_digests(N, capl) * 4))) return rc;
    // witness tree; the leaf kernel (the dominant kernel of a proof) is bracketed by HIP events on the launch stream (with a
    // host witness hashed in groups the bracket spans the groups' transforms too: the roofline figure comes from bj_prove_dev)
    if (!hashed_in_groups) {
        BJ_HIP(ctx, hipEventRecord(ctx->ev0, st));
        bj::launch_tree_leaves(ctx->hasher, wit_lde.p, Ln, nullptr, nW, N, wit_tree.p, st);
        BJ_HIP(ctx, hipEventRecord(ctx->ev1, st));
    }
    bj::launch_tree_node_layers(ctx->hasher, wit_tree.p, N, capl, st);
    BJ_CHECK_LAUNCH(ctx);
    std::vector<u64> wit_cap(4 * cap), s2_cap(4 * cap), q_cap(4 * cap);
    rc = bj::gather_cap(ctx, sh, wit_tree.p, N, cap, wit_cap.data());
    if (rc) return rc;
    BJ_HIP(ctx, hipEventElapsedTime(&proof->stage_ms[7], ctx->ev0, ctx->ev1));
    tr.absorb_cap(wit_cap.data(), wit_cap.size());
    proof->stage_ms[0] = timer.lap();

    // ---------------- round 2: copy-permutation + lookup polys (prover.rs:360-554) ----------------
    u64 beta[2], gamma[2], lbeta[2] = {0, 0}, lgamma[2] = {0, 0};
    challenge2(beta);
    challenge2(gamma);
    ArenaBuf s2_nat, s2_lde, s2_tree, tmp;
    if ((rc = s2_nat.alloc(ctx, (size_t)nS2 * n))) return rc;
    if ((rc = tmp.alloc(ctx, (size_t)2 * n_chunks * n + 2 * ((n + 1023) / 1024) + 16))) return rc;
    const u64 *d_sig_nat = S->d_nat, *d_con_nat = S->d_nat + (size_t)V * n, *d_tab_nat = S->d_nat + (size_t)(V + nC) * n;
    bj::launch_copy_perm_stage2(d_variables, n, d_sig_nat, n, S->d_non_res, V, q, log_n, ctx->tw_fwd, beta, gamma, tmp.p,
                                s2_nat.p, s2_nat.p + 2 * n, st, S->small_non_residues);
    if (has_lookup) {
        challenge2(lbeta);
        challenge2(lgamma);
        u64 *dA = s2_nat.p + (size_t)(2 + 2 * n_part) * n, *dB = dA + (size_t)2 * S->lookup_reps * n;
        bj::launch_lookup_polys(d_variables + (size_t)S->num_gp_vars * n, n, S->tid_var ? nullptr : d_con_nat + (size_t)S->table_id_col * n, d_tab_nat,
                                n, d_multiplicities, S->lookup_reps, S->lookup_w, log_n, lbeta, lgamma, dA, dB, st);
    }
    BJ_CHECK_LAUNCH(ctx);
    if ((rc = s2_lde.alloc(ctx, (size_t)nS2 * Ln))) return rc;
    rc = intt_cols(s2_nat.p, mono_s2.p, nS2);
    if (!rc) rc = lde_cols(mono_s2.p, s2_lde.p, Ln, nS2, S->log_L, S->c0, S->cl);
    if (rc) return rc;
    if ((rc = s2_tree.alloc(ctx, bj_merkle_tree_digests(N, capl) * 4))) return rc;
    rc = bj_merkle_tree_build(ctx, s2_lde.p, Ln, nS2, N, capl, s2_tree.p);
    if (!rc) rc = bj::gather_cap(ctx, sh, s2_tree.p, N, cap, s2_cap.data());
    if (rc) return rc;
    tr.absorb_cap(s2_cap.data(), s2_cap.size());
    proof->stage_ms[1] = timer.lap();

    // ---------------- round 3: quotient (prover.rs:560-1495) ----------------
    u64 alpha[2];
    challenge2(alpha);
    const unsigned n_lookup_terms = has_lookup ? S->lookup_reps + 1 : 0;
    unsigned n_gate_terms = 0;
    for (unsigned g = 0; g < S->n_gates; g++) n_gate_terms += (unsigned)(S->gates_flat[12 * g + 2] * S->gates_flat[12 * g + 5]);
    const unsigned n_spec_terms = S->n_spec_terms;   // lookup | specialized | general | L1 | copy-permutation (prover.rs:599-625)
    const unsigned total_terms = n_lookup_terms + n_spec_terms + n_gate_terms + 1 + n_chunks;
    std::vector<u64> alphas(2 * total_terms);
    {
        gl::e2 a = e2c(alpha), cur{1, 0};
        for (unsigned i = 0; i < total_terms; i++) {   // materialize_powers_serial (utils.rs:31)
            alphas[2 * i] = cur.c0;
            alphas[2 * i + 1] = cur.c1;
            cur = gl::e2_mul(cur, a);
        }
    }
    ArenaBuf d_alphas, T;
    if ((rc = d_alphas.alloc(ctx, alphas.size()))) return rc;
    if ((rc = bj::h2d_async(ctx, d_alphas.p, alphas.data(), alphas.size() * 8))) return rc;
    if ((rc = T.alloc(ctx, 2 * Q))) return rc;
    ArenaBuf Tl;   // this rank's evaluations [2][Qe] (T itself when nothing has to be gathered)
    if (q_local)
        Tl.p = T.p;
    else if ((rc = Tl.alloc(ctx, 2 * Qe))) return rc;
    u64 *t0 = Tl.p, *t1 = Tl.p + Qe;
    const u64 *a_lookup = d_alphas.p, *a_spec = d_alphas.p + 2 * n_lookup_terms, *a_gates = a_spec + 2 * n_spec_terms,
              *a_l1 = a_gates + 2 * n_gate_terms;
    const u64 *d_sig_lde = S->d_lde, *d_con_lde = S->d_lde + (size_t)V * Ln, *d_tab_lde = S->d_lde + (size_t)(V + nC) * Ln;
    if (Qe) {
        // gate evaluation, SURVEY §8d: 8 (V + consts) qn + 16 qn — V = the general-purpose columns the gates read
        const int pg = bj::probe_begin(ctx, "quotient_gates", 8.0 * (S->num_gp_vars + nC) * (double)Qe + 16.0 * (double)Qe);
        bj::launch_quotient_gates(wit_lde.p, Ln, d_con_lde, Ln, S->gates_flat.data(), S->n_gates, a_gates, Qe, t0, t1, st);
        bj::probe_end(ctx, pg);
        unsigned aoff = 0;   // op-list gates (seam S3) add their contribution on top, with their own slice of alpha powers
        std::vector<bj::GateLaunch> prog_gates;
        for (unsigned g = 0; g < S->n_gates; g++) {
            const int *f = S->gates_flat.data() + 12 * g;
            if (f[0] == BJ_GATE_POSEIDON2_FLATTENED) {
                unsigned char path[8] = {0};
                for (int b = 0; b < f[1]; b++) path[b] = (unsigned char)f[6 + b];
                bj::launch_quotient_poseidon2_flattened(wit_lde.p, Ln, d_con_lde, Ln, (unsigned)f[1], path, a_gates + 2 * (size_t)aoff,
                                                        Qe, t0, t1, st);
            }
            if (f[0] == BJ_GATE_PROGRAM) {
                bj::GateLaunch L{};
                L.program = &S->programs[g];
                L.path_len = (unsigned)f[1];
                for (int b = 0; b < f[1]; b++) L.path[b] = (unsigned char)f[6 + b];
                L.reps = (unsigned)f[2];
                L.rep_var_stride = (unsigned)f[3];
                L.rep_const_stride = (unsigned)f[4];
                L.rep_wit_stride = S->gate_wit_stride[g];
                L.d_alphas = a_gates + 2 * (size_t)aoff;
                prog_gates.push_back(L);
            }
            aoff += (unsigned)(f[2] * f[5]);
        }
        // the op-list gates: one fused launch for those with generated bodies (they all sweep the general-purpose columns)
        bj::launch_gate_programs(prog_gates.data(), (unsigned)prog_gates.size(), wit_lde.p, Ln, d_con_lde, Ln, Qe, t0, t1, st,
                                 S->Wc ? wit_lde.p + (size_t)V * Ln : nullptr);
        unsigned soff = 0;   // gates over specialized columns: every row, no selector
        const unsigned char no_path[8] = {0};
        for (const auto &sg : S->spec) {
            bj::launch_gate_program(sg.program, wit_lde.p + (size_t)sg.first_col * Ln, Ln, d_con_lde + (size_t)sg.first_const * Ln, Ln, 0,
                                    no_path, sg.reps, sg.width, sg.const_width, a_spec + 2 * (size_t)soff, Qe, t0, t1, nullptr, st);
            soff += sg.reps * sg.terms;
        }
    }
    if (has_lookup && Qe) {
        const u64 *dA = s2_lde.p + (size_t)(2 + 2 * n_part) * Ln, *dB = dA + (size_t)2 * S->lookup_reps * Ln;
        bj::launch_quotient_lookup(wit_lde.p + (size_t)S->num_gp_vars * Ln, Ln, S->tid_var ? nullptr : d_con_lde + (size_t)S->table_id_col * Ln, d_tab_lde,
                                   Ln, wit_lde.p + (size_t)VW * Ln, dA, dB, Ln, S->lookup_reps, S->lookup_w, lbeta, lgamma,
                                   a_lookup, Qe, t0, t1, st);
    }
    if (Qe) {
        // variables + sigmas + z and the partial products + 1/(x - 1), accumulators read and written
        const int pc = bj::probe_begin(ctx, "quotient_copy_perm", 8.0 * (2.0 * V + 2 + 2 * n_part + 1) * (double)Qe + 32.0 * (double)Qe);
        bj::launch_quotient_copy_perm(wit_lde.p, Ln, d_sig_lde, Ln, s2_lde.p, Ln, S->d_non_res, V, q, log_n, S->log_L, ctx->tw_fwd,
                                      beta, gamma, alphas.data() + 2 * (n_lookup_terms + n_spec_terms + n_gate_terms), a_l1 + 2, Qe, I0, S->d_inv_xm1, t0, t1, st,
                                      S->small_non_residues);
        bj::probe_end(ctx, pc);
    }
    BJ_CHECK_LAUNCH(ctx);
    // flatten (= bit-reversal of the evaluations), iNTT on their coset (prover.rs:1386-1422)
    const unsigned log_Q = log_n + S->log_q;
    if (q_local) {
        rc = bj_bitreverse_batch(ctx, T.p, T.p, log_Q, 2, Q);
        if (!rc) rc = bj_intt_batch(ctx, T.p, T.p, log_Q, 2, Q, gl::GEN);
    } else {
        // Sharded: the first Qe points of this rank are s * H_Qe with s = x_{I0} = 7 * w_{Ln}^{bitrev(c0)}: the inverse transform
        // with that shift gives R = T mod (x^Qe - a), a = s^Qe.  With T = sum_j x^(j Qe) T_j, R_i = sum_j a_i^j T_j: W residues
        // (all-gathered: 2 Qe words from every rank, 2 q n in total) determine T by a W x W Vandermonde solve per coefficient
        // (combine_residues_kernel) — no rank evaluates a point twice and the size-q n inverse transform is not replicated.
        const unsigned log_E = bj::log2_exact(Qe);
        auto rank_shift = [&](unsigned r) {
            return gl::mul(gl::GEN, gl::pow(gl::omega(log_n + S->log_L), gl::bitrev32(r * S->cl, S->log_L)));
        };
        rc = bj_bitreverse_batch(ctx, Tl.p, Tl.p, log_E, 2, Qe);
        if (!rc) rc = bj_intt_batch(ctx, Tl.p, Tl.p, log_E, 2, Qe, rank_shift(sh.rank));
        if (rc) return rc;
        ArenaBuf all;
        if ((rc = all.alloc(ctx, (size_t)2 * Q))) return rc;
        if ((rc = bj::all_gather(ctx, sh, Tl.p, all.p, 2 * Qe))) return rc;
        u64 a[8];
        for (unsigned r = 0; r < sh.world; r++) a[r] = gl::pow(rank_shift(r), Qe);
        if (!bj::launch_combine_residues(all.p, sh.world, Qe, 2, a, T.p, st))
            return bj::fail(ctx, BJ_ERR_INVALID_ARG, "sharded quotient: the residues' moduli are not distinct");
        BJ_CHECK_LAUNCH(ctx);
    }
    u64 top[2] = {1, 1};
    if (!rc) rc = bj_memcpy_d2h(ctx, &top[0], T.p + Q - 1, 8);
    if (!rc) rc = bj_memcpy_d2h(ctx, &top[1], T.p + 2 * Q - 1, 8);
    if (rc) return rc;
    if (top[0] != 0 || top[1] != 0)
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_prove: constraint system is not satisfied (quotient is not a polynomial; prover.rs:1425-1438)");
    ArenaBuf q_lde, q_tree, Tt;
    const u64 *Tm = T.p;   // the 2 q chunks of n coefficients each, in the monomial layout of this trace length
    if (tiled) {           // they come out of a transform of another size: one re-layout pass over 2 q n words
        if ((rc = Tt.alloc(ctx, 2 * Q))) return rc;
        bj::launch_tiled_permute(T.p, Tt.p, 2 * q, n, n, true, st);
        BJ_CHECK_LAUNCH(ctx);
        Tm = Tt.p;
    }
    if ((rc = q_lde.alloc(ctx, (size_t)2 * q * N))) return rc;
    for (unsigned e = 0; e < 2 && !rc; e++)   // chunk j of c_e -> column 2j+e (prover.rs:1445-1467)
        rc = lde_cols(Tm + (size_t)e * Q, q_lde.p + (size_t)e * N, 2 * N, q, S->log_fri, sh.world > 1 ? S->c0 : 0, sh.world > 1 ? S->cl : fri);
    if (rc) return rc;
    if ((rc = q_tree.alloc(ctx, bj_merkle_tree_digests(N, capl) * 4))) return rc;
    rc = bj_merkle_tree_build(ctx, q_lde.p, N, 2 * q, N, capl, q_tree.p);
    if (!rc) rc = bj::gather_cap(ctx, sh, q_tree.p, N, cap, q_cap.data());
    if (rc) return rc;
    tr.absorb_cap(q_cap.data(), q_cap.size());
    proof->stage_ms[2] = timer.lap();

    // ---------------- round 4: openings (prover.rs:1501-1802) ----------------
    u64 z[2];
    challenge2(z);
    ArenaBuf w;
    if ((rc = w.alloc(ctx, 2 * n))) return rc;
    // base columns whose coset 0 is evaluated, in the order of prover.rs:1550-1683; F_p^2 polys contribute two columns
    struct Src { const u64 *c0, *c1; };
    std::vector<Src> srcs, msrcs;   // msrcs: the monomial forms of the same polynomials, same order (for the DEEP numerator)
    const u64 *m_sig = S->d_mono, *m_con = S->d_mono + (size_t)V * n, *m_tab = S->d_mono + (size_t)(V + nC) * n;
    for (unsigned i = 0; i < VW; i++) srcs.push_back({wit_lde.p + (size_t)i * Ln, nullptr}), msrcs.push_back({mono.p + (size_t)i * n, nullptr});   // variables, witness
    for (unsigned i = 0; i < nC; i++) srcs.push_back({d_con_lde + (size_t)i * Ln, nullptr}), msrcs.push_back({m_con + (size_t)i * n, nullptr});
    for (unsigned i = 0; i < V; i++) srcs.push_back({d_sig_lde + (size_t)i * Ln, nullptr}), msrcs.push_back({m_sig + (size_t)i * n, nullptr});
    for (unsigned j = 0; j < 1 + n_part; j++) {
        srcs.push_back({s2_lde.p + (size_t)(2 * j) * Ln, s2_lde.p + (size_t)(2 * j + 1) * Ln});
        msrcs.push_back({mono_s2.p + (size_t)(2 * j) * n, mono_s2.p + (size_t)(2 * j + 1) * n});
    }
    if (has_lookup) {
        srcs.push_back({wit_lde.p + (size_t)VW * Ln, nullptr});
        msrcs.push_back({mono.p + (size_t)VW * n, nullptr});
        for (unsigned i = 0; i < S->lookup_reps + 1; i++) {
            size_t o = (size_t)(2 + 2 * n_part + 2 * i);
            srcs.push_back({s2_lde.p + o * Ln, s2_lde.p + (o + 1) * Ln});
            msrcs.push_back({mono_s2.p + o * n, mono_s2.p + (o + 1) * n});
        }
        for (unsigned i = 0; i < nT; i++) srcs.push_back({d_tab_lde + (size_t)i * Ln, nullptr}), msrcs.push_back({m_tab + (size_t)i * n, nullptr});
    }
    for (unsigned j = 0; j < q; j++) {   // chunk j of the quotient: coefficients [j*n, (j+1)*n) of T's monomial form
        srcs.push_back({q_lde.p + (size_t)(2 * j) * N, q_lde.p + (size_t)(2 * j + 1) * N});
        msrcs.push_back({Tm + (size_t)j * n, Tm + Q + (size_t)j * n});
    }
    // every rank evaluates from ITS first coset (shift 7*w^bitrev(c0)): the polynomials have degree < n, so the value
    // is the same field element whichever coset it is interpolated from — no exchange, identical transcripts
    const u64 open_shift = gl::mul(gl::GEN, gl::pow(gl::omega(log_n + S->log_L), gl::bitrev32(S->c0, S->log_L)));
    auto evaluate = [&](const std::vector<Src> &ss, const u64 *at, std::vector<u64> &vals) -> int {
        int r = bj_barycentric_weights(ctx, log_n, open_shift, at, w.p, w.p + n);
        if (r) return r;
        std::vector<const u64 *> ptrs;
        for (auto &s : ss) {
            ptrs.push_back(s.c0);
            if (s.c1) ptrs.push_back(s.c1);
        }
        std::vector<u64> raw(2 * ptrs.size());
        const size_t P = ptrs.size(), per = (P + sh.world - 1) / sh.world;
        if (sh.world > 1 && P >= 8 * (size_t)sh.world && 2 * per * sh.world <= 2048) {
            // many columns: rank r evaluates the slice [r*per, (r+1)*per) on its own coset, the values are all-gathered
            // (the value of a polynomial does not depend on the coset it was interpolated from)
            const size_t lo = (size_t)sh.rank * per, hi = lo + per < P ? lo + per : P;
            std::vector<u64> mine(2 * per, 0);
            if (hi > lo) {
                r = bj_barycentric_eval_batch(ctx, ptrs.data() + lo, (unsigned)(hi - lo), log_n, w.p, w.p + n, mine.data());
                if (r) return r;
            }
            u64 *d_part = ctx->d_small + 64 + 64 * 32, *d_all = d_part + 2048;   // the 4096-u64 gather area of the context
            if ((r = bj::h2d_async(ctx, d_part, mine.data(), 2 * per * 8))) return r;
            if ((r = bj::all_gather(ctx, sh, d_part, d_all, 2 * per))) return r;
            std::vector<u64> all(2 * per * sh.world);
            if ((r = bj_memcpy_d2h(ctx, all.data(), d_all, all.size() * 8))) return r;
            std::memcpy(raw.data(), all.data(), raw.size() * 8);
        } else {
            r = bj_barycentric_eval_batch(ctx, ptrs.data(), (unsigned)P, log_n, w.p, w.p + n, raw.data());
            if (r) return r;
        }
        vals.clear();
        size_t k = 0;
        for (auto &s : ss) {
            gl::e2 e0{raw[2 * k], raw[2 * k + 1]};
            k++;
            if (s.c1) {   // E(c0) + u * E(c1) = (a0 + 7 b1, a1 + b0)
                gl::e2 e1{raw[2 * k], raw[2 * k + 1]};
                k++;
                e0 = {gl::add(e0.c0, gl::mul(gl::GEN, e1.c1)), gl::add(e0.c1, e1.c0)};
            }
            vals.push_back(e0.c0);
            vals.push_back(e0.c1);
        }
        return BJ_OK;
    };
    std::vector<u64> vz, vzo, v0;
    if ((rc = evaluate(srcs, z, vz))) return rc;
    tr.absorb(vz.data(), vz.size());
    u64 zo[2];
    {
        u64 om = gl::omega(log_n);
        zo[0] = gl::mul(gl::canon(z[0]), om);
        zo[1] = gl::mul(gl::canon(z[1]), om);
    }
    std::vector<Src> zsrc{srcs[VW + nC + V]};
    if ((rc = evaluate(zsrc, zo, vzo))) return rc;
    tr.absorb(vzo.data(), vzo.size());
    std::vector<Src> lsrc;
    if (has_lookup) {
        for (unsigned i = 0; i < S->lookup_reps + 1; i++) lsrc.push_back(srcs[VW + nC + V + 1 + n_part + 1 + i]);
        u64 zero2[2] = {0, 0};
        if ((rc = evaluate(lsrc, zero2, v0))) return rc;
        tr.absorb(v0.data(), v0.size());
    }
    proof->stage_ms[3] = timer.lap();

    // ---------------- round 5a: DEEP (prover.rs:1803-2067) ----------------
    struct PubSet { u64 at; std::vector<unsigned> cols; std::vector<u64> vals; };
    std::vector<PubSet> pubs;
    {
        u64 om = gl::omega(log_n);
        for (size_t i = 0; i < S->pub_cols.size(); i++) {
            u64 at = gl::pow(om, S->pub_rows[i]);
            size_t pos = 0;
            for (; pos < pubs.size(); pos++)
                if (pubs[pos].at == at) break;
            if (pos == pubs.size()) pubs.push_back({at, {}, {}});
            pubs[pos].cols.push_back(S->pub_cols[i]);
            pubs[pos].vals.push_back(gl::canon(h_public_values[i]));
        }
    }
    u64 cch[2];
    challenge2(cch);
    size_t total_ch = srcs.size() + 1 + lsrc.size();
    for (auto &p : pubs) total_ch += p.cols.size();
    std::vector<u64> chs(2 * total_ch);
    {
        gl::e2 c = e2c(cch), cur{1, 0};
        for (size_t i = 0; i < total_ch; i++) {   // materialize_ext_challenge_powers (prover.rs:2374-2395)
            chs[2 * i] = cur.c0;
            chs[2 * i + 1] = cur.c1;
            cur = gl::e2_mul(cur, c);
        }
    }
    ArenaBuf deep;
    if ((rc = deep.alloc(ctx, 2 * N))) return rc;
    size_t choff = 0;
    // Every opening set goes the same way: the numerator sum_k ch_k f_k is a polynomial of degree < n, so it is combined on
    // the MONOMIAL forms (n coefficients per column instead of the fri_lde_factor * n values of the FRI domain), extended by
    // one two-column LDE and divided by (x - at) pointwise.  Exact arithmetic: the same values as combining on the LDE.
    ArenaBuf num_mono, num_slice;
    if ((rc = num_mono.alloc(ctx, 2 * n))) return rc;
    if (sh.world > 1 && (rc = num_slice.alloc(ctx, 2 * n / sh.world + 16))) return rc;
    // The sets are prepared one after the other (a large one leaves its extended numerator in a buffer of its own) and divided
    // by their (x - at) TOGETHER, bj::DEEP_MAX_SETS per launch: one inversion per lane for all of them, the destination written once.
    struct PendingDeep {
        std::vector<const u64 *> p0, p1;
        std::vector<u64> vals, ch;
        u64 at[2];
    };
    std::deque<PendingDeep> pending;
    bool deep_written = false;
    auto flush_deep = [&]() -> int {
        while (!pending.empty()) {
            bj::DeepSetHost hs[bj::DEEP_MAX_SETS];
            unsigned cnt = 0;
            for (auto it = pending.begin(); it != pending.end() && cnt < (unsigned)bj::DEEP_MAX_SETS; ++it, ++cnt)
                hs[cnt] = bj::DeepSetHost{it->p0.data(), it->p1.data(), it->p0.size(), it->vals.data(), it->ch.data(), it->at};
            double deep_cols = 0;      // DEEP, SURVEY §8d: 8 (#base columns) Ln + 16 Ln (the destination pair)
            for (unsigned k = 0; k < cnt; k++)
                for (size_t j = 0; j < hs[k].n_src; j++) deep_cols += hs[k].src_c1 && hs[k].src_c1[j] ? 2 : 1;
            const int pd = bj::probe_begin(ctx, "deep_accumulate_multi", 8.0 * deep_cols * (double)N + (deep_written ? 32.0 : 16.0) * (double)N);
            int r = bj::deep_accumulate_multi(ctx, hs, cnt, log_n, S->log_fri, N, sh.world > 1 ? I0 : 0, deep.p, deep.p + N,
                                              deep_written ? 1 : 0);
            bj::probe_end(ctx, pd);
            if (r) return r;
            deep_written = true;
            for (unsigned k = 0; k < cnt; k++) pending.pop_front();
        }
        return BJ_OK;
    };
    auto deep_set = [&](const std::vector<Src> &ls, const std::vector<Src> &ms, const u64 *vals, const u64 *at) -> int {
        const u64 *ch = chs.data() + 2 * choff;
        std::vector<const u64 *> p0, p1;
        size_t n_base = 0;
        for (auto &m : ms) n_base += m.c1 ? 2 : 1;
        if (n_base < 16) {   // a handful of columns: streaming them over the FRI domain is cheaper than an extra LDE pass
            PendingDeep pd;
            for (auto &l : ls) {
                pd.p0.push_back(l.c0);
                pd.p1.push_back(l.c1);
            }
            pd.vals.assign(vals, vals + 2 * ls.size());
            pd.ch.assign(ch, ch + 2 * ls.size());
            pd.at[0] = at[0];
            pd.at[1] = at[1];
            pending.push_back(std::move(pd));
            choff += ls.size();
            return BJ_OK;
        }
        ArenaBuf num_lde;   // this set's extended numerator: alive until the sets are flushed
        if (int ra = num_lde.alloc(ctx, 2 * N)) return ra;
        for (auto &m : ms) {
            p0.push_back(m.c0);
            p1.push_back(m.c1);
        }
        int r;
        if (sh.world > 1 && n % sh.world == 0 && n / sh.world >= 256) {
            // the monomials are replicated: every rank combines its slice of the coefficient range, one all-gather of the
            // two result columns rebuilds the numerator everywhere (the combination is the replicated part of DEEP)
            const size_t per = n / sh.world, off = (size_t)sh.rank * per;
            for (auto &q0 : p0) q0 += off;
            for (auto &q1 : p1)
                if (q1) q1 += off;
            r = bj::combine_monomials(ctx, p0.data(), p1.data(), ms.size(), ch, per, num_slice.p, num_slice.p + per);
            if (!r) r = bj::all_gather_columns(ctx, sh, num_slice.p, num_mono.p, 2, per);
        } else {
            r = bj::combine_monomials(ctx, p0.data(), p1.data(), ms.size(), ch, n, num_mono.p, num_mono.p + n);
        }
        if (r) return r;
        if (sh.world > 1)
            r = lde_cols(num_mono.p, num_lde.p, (size_t)S->cl << log_n, 2, S->log_L, S->c0, S->cl);
        else
            r = lde_cols(num_mono.p, num_lde.p, N, 2, S->log_fri, 0, fri);
        if (r) return r;
        gl::e2 C{0, 0};   // sum_k ch_k * v_k
        for (size_t k = 0; k < ms.size(); k++) C = gl::e2_add(C, gl::e2_mul(e2c(ch + 2 * k), e2c(vals + 2 * k)));
        PendingDeep pd;   // one F_p^2 source (the extended numerator) with challenge 1 and "value" C
        pd.p0.push_back(num_lde.p);
        pd.p1.push_back(num_lde.p + N);
        pd.vals = {C.c0, C.c1};
        pd.ch = {1, 0};
        pd.at[0] = at[0];
        pd.at[1] = at[1];
        pending.push_back(std::move(pd));
        choff += ms.size();
        return BJ_OK;
    };
    if ((rc = deep_set(srcs, msrcs, vz.data(), z))) return rc;
    {
        std::vector<Src> mz{msrcs[VW + nC + V]};                      // z(x) at z*omega
        if ((rc = deep_set(zsrc, mz, vzo.data(), zo))) return rc;
    }
    if (has_lookup) {
        std::vector<Src> ml;
        for (unsigned i = 0; i < S->lookup_reps + 1; i++) ml.push_back(msrcs[VW + nC + V + 1 + n_part + 1 + i]);
        u64 zero2[2] = {0, 0};
        if ((rc = deep_set(lsrc, ml, v0.data(), zero2))) return rc;
    }
    for (auto &p : pubs) {
        std::vector<Src> ps, pl;
        std::vector<u64> pv;
        for (size_t i = 0; i < p.cols.size(); i++) {
            pl.push_back({wit_lde.p + (size_t)p.cols[i] * Ln, nullptr});
            ps.push_back({mono.p + (size_t)p.cols[i] * n, nullptr});
            pv.push_back(p.vals[i]);
            pv.push_back(0);
        }
        u64 at2[2] = {p.at, 0};
        if ((rc = deep_set(pl, ps, pv.data(), at2))) return rc;
    }
    if ((rc = flush_deep())) return rc;
    proof->stage_ms[4] = timer.lap();

    // ---------------- round 5b: FRI (prover.rs:2075-2105) ----------------
    uint32_t sched[32], new_pow = 0;
    size_t sched_len = 0, num_queries = 0, final_degree = 0;
    if ((rc = bj_fri_schedule(S->security, cap, S->pow_bits, S->log_fri, log_n, &new_pow, &num_queries, sched, &sched_len, &final_degree)))
        return bj::fail(ctx, rc, "bj_prove: compute_fri_schedule failed");
    bj_transcript trw;   // bj_fri_prove drives a bj_transcript; hand our state over and take it back
    trw.t = tr;
    bj_fri *fri_obj = nullptr;
    rc = bj::fri_prove_sharded(ctx, sh, deep.p, deep.p + N, log_n, S->log_fri, sched, sched_len, cap, &trw, &fri_obj);
    if (rc) return rc;
    struct FriGuard {
        bj_fri *f;
        ~FriGuard() { bj_fri_destroy(f); }
    } fri_guard{fri_obj};
    tr = trw.t;
    // ---------------- proof of work (prover.rs:2107-2131; PoWRunner = Blake2s256, pow.rs:50-133, or Keccak256, pow.rs:139-230) ----------------
    u64 pow_challenge = 0;
    if (new_pow) {
        u64 seed[5];   // 256 / CHAR_BITS = 4, "+1 if not a multiple of CHAR_BITS" -> 5 challenges = 40 seed bytes
        for (int i = 0; i < 5; i++) seed[i] = gl::canon(tr.challenge());
        ArenaBuf d_res;
        if ((rc = d_res.alloc(ctx, 8))) return rc;
        const u64 none = ~(u64)0, batch = (u64)1 << 24;
        u64 found = none;
        for (u64 base = 0; found == none; base += batch) {   // batches in order + minimum inside a batch = the smallest nonce,
            if ((rc = bj::h2d_async(ctx, d_res.p, &none, 8))) return rc;   // i.e. what the reference's serial search returns
            if (S->pow_runner == BJ_POW_KECCAK256) bj::launch_keccak_pow(seed, new_pow, base, batch, d_res.p, st);
            else bj::launch_blake2s_pow(seed, new_pow, base, batch, d_res.p, st);
            BJ_CHECK_LAUNCH(ctx);
            if ((rc = bj_memcpy_d2h(ctx, &found, d_res.p, 8))) return rc;
            if (base > ((u64)1 << 40)) return bj::fail(ctx, BJ_ERR_HIP, "bj_prove: proof of work did not terminate");
        }
        pow_challenge = found;
        const u64 lh[2] = {found & 0xFFFFFFFFULL, found >> 32};
        tr.absorb(lh, 2);
    }
    proof->stage_ms[5] = timer.lap();

    // ---------------- round 6: queries (prover.rs:2161-2266) ----------------
    bj::host::BoolsBuffer bools;
    bools.max_needed = log_n + S->log_fri;
    std::vector<u64> idxs(num_queries);
    for (size_t i = 0; i < num_queries; i++) idxs[i] = bools.query_index(tr, log_n, S->log_fri);
    const unsigned depth = bj::log2_exact(N / capl);
    const unsigned W = sh.world;
    // a leaf lives on rank index / N; every rank gathers at (index mod N) and the owner's answer is kept
    const unsigned widths[4] = {nW, nS2, 2 * q, S->n_cols};
    const u64 *bases[4] = {wit_lde.p, s2_lde.p, q_lde.p, S->d_lde};
    const size_t strides[4] = {Ln, Ln, N, Ln};
    const u64 *trees[4] = {wit_tree.p, s2_tree.p, q_tree.p, S->d_tree};
    ArenaBuf d_idx, d_g;
    size_t per_query = 0;
    for (int o = 0; o < 4; o++) per_query += widths[o] + (size_t)depth * 4;
    if ((rc = d_idx.alloc(ctx, num_queries))) return rc;
    if ((rc = d_g.alloc(ctx, per_query * num_queries))) return rc;
    {
        std::vector<u64> loc(num_queries);
        for (size_t i = 0; i < num_queries; i++) loc[i] = idxs[i] % N;
        if ((rc = bj::h2d_async(ctx, d_idx.p, loc.data(), num_queries * 8))) return rc;
    }
    const size_t G = per_query * num_queries;
    std::vector<u64> gathered(G * W);   // [rank][...]; query qi reads the block of rank idxs[qi] / N
    {
        size_t off = 0;
        for (int o = 0; o < 4; o++) {
            bj::launch_gather_rows(bases[o], strides[o], widths[o], d_idx.p, (unsigned)num_queries, d_g.p + off, st);
            off += (size_t)widths[o] * num_queries;
            bj::launch_merkle_paths(trees[o], N, depth, d_idx.p, (unsigned)num_queries, d_g.p + off, st);
            off += (size_t)depth * 4 * num_queries;
        }
        BJ_CHECK_LAUNCH(ctx);
        const u64 *src = d_g.p;
        ArenaBuf all;
        if (W > 1) {
            if ((rc = all.alloc(ctx, G * W))) return rc;
            if ((rc = bj::all_gather(ctx, sh, d_g.p, all.p, G))) return rc;
            src = all.p;
        }
        if ((rc = bj_memcpy_d2h(ctx, gathered.data(), src, gathered.size() * 8))) return rc;
    }
    // FRI openings, batched per oracle: leaf j = (index >> folds so far) >> k  (proof.rs:65-100, fri/mod.rs:829-895)
    std::vector<std::vector<u64>> fri_leaves(sched_len), fri_paths(sched_len);
    std::vector<unsigned> fri_depth(sched_len);
    {
        ArenaBuf d_li, d_fo;
        if ((rc = d_li.alloc(ctx, num_queries))) return rc;
        size_t max_out = 0;
        for (size_t i = 0; i < sched_len; i++) {
            const bj_fri::Oracle &o = fri_obj->oracles[i];
            fri_depth[i] = bj::log2_exact(o.num_leaves / (cap / o.world));
            size_t need = ((size_t)2 << o.log_e) + (size_t)fri_depth[i] * 4;
            if (need > max_out) max_out = need;
        }
        if ((rc = d_fo.alloc(ctx, max_out * num_queries * (W + 1)))) return rc;
        std::vector<u64> li(num_queries), host_all;
        unsigned shift = 0;
        for (size_t i = 0; i < sched_len; i++) {
            const bj_fri::Oracle &o = fri_obj->oracles[i];
            for (size_t qi = 0; qi < num_queries; qi++) li[qi] = ((idxs[qi] >> shift) >> o.log_e) % o.num_leaves;
            const unsigned shift0 = shift;
            shift += o.log_e;
            if ((rc = bj::h2d_async(ctx, d_li.p, li.data(), num_queries * 8))) return rc;
            const size_t E2 = (size_t)2 << o.log_e;
            bj::launch_gather_fri_leaves(o.d_c0, o.d_c1, o.log_e, d_li.p, (unsigned)num_queries, d_fo.p, st);
            bj::launch_merkle_paths(o.d_tree, o.num_leaves, fri_depth[i], d_li.p, (unsigned)num_queries,
                                    d_fo.p + E2 * num_queries, st);
            BJ_CHECK_LAUNCH(ctx);
            fri_leaves[i].resize(E2 * num_queries);
            fri_paths[i].resize((size_t)fri_depth[i] * 4 * num_queries + 1);
            const size_t PD = (size_t)fri_depth[i] * 4, blk = (E2 + PD) * num_queries;
            if (o.world > 1) {   // oracle 0 of a sharded proof: keep the owner's leaf and path
                u64 *d_all = d_fo.p + max_out * num_queries;
                if ((rc = bj::all_gather(ctx, sh, d_fo.p, d_all, blk))) return rc;
                host_all.resize(blk * W);
                if ((rc = bj_memcpy_d2h(ctx, host_all.data(), d_all, blk * W * 8))) return rc;
                for (size_t qi = 0; qi < num_queries; qi++) {
                    const size_t owner = ((idxs[qi] >> shift0) >> o.log_e) / o.num_leaves;
                    const u64 *b = host_all.data() + owner * blk;
                    std::memcpy(fri_leaves[i].data() + qi * E2, b + qi * E2, E2 * 8);
                    std::memcpy(fri_paths[i].data() + qi * PD, b + E2 * num_queries + qi * PD, PD * 8);
                }
                continue;
            }
            if ((rc = bj_memcpy_d2h(ctx, fri_leaves[i].data(), d_fo.p, E2 * num_queries * 8))) return rc;
            if (fri_depth[i] &&
                (rc = bj_memcpy_d2h(ctx, fri_paths[i].data(), d_fo.p + E2 * num_queries, (size_t)fri_depth[i] * 4 * num_queries * 8)))
                return rc;
        }
    }
    // ---------------- serialise ----------------
    std::vector<u64> &D = proof->data;
    auto put = [&](const u64 *p, size_t k) { D.insert(D.end(), p, p + k); };
    const u64 header[] = {0x424A5046ULL, 2, S->pub_cols.size(), cap, vz.size() / 2, vzo.size() / 2, v0.size() / 2, sched_len,
                          final_degree, num_queries, nW, nS2, 2 * q, S->n_cols, depth, log_n, fri, S->pow_bits, pow_challenge};
    put(header, sizeof(header) / 8);
    for (size_t i = 0; i < sched_len; i++) D.push_back(sched[i]);
    for (size_t i = 0; i < S->pub_cols.size(); i++) D.push_back(gl::canon(h_public_values[i]));
    put(wit_cap.data(), wit_cap.size());
    put(s2_cap.data(), s2_cap.size());
    put(q_cap.data(), q_cap.size());
    put(vz.data(), vz.size());
    put(vzo.data(), vzo.size());
    put(v0.data(), v0.size());
    {
        std::vector<u64> c(4 * cap);
        for (size_t i = 0; i < sched_len; i++) {
            bj_fri_cap(fri_obj, i, c.data());
            put(c.data(), c.size());
        }
        std::vector<u64> f0(final_degree), f1(final_degree);
        bj_fri_final_monomials(fri_obj, f0.data(), f1.data());
        put(f0.data(), final_degree);
        put(f1.data(), final_degree);
    }
    for (size_t qi = 0; qi < num_queries; qi++) {
        D.push_back(idxs[qi]);
        size_t off = (idxs[qi] / N) * G;
        for (int o = 0; o < 4; o++) {
            put(gathered.data() + off + qi * widths[o], widths[o]);
            off += (size_t)widths[o] * num_queries;
            put(gathered.data() + off + qi * (size_t)depth * 4, (size_t)depth * 4);
            off += (size_t)depth * 4 * num_queries;
        }
        for (size_t i = 0; i < sched_len; i++) {
            const size_t E2 = (size_t)2 << sched[i];
            put(fri_leaves[i].data() + qi * E2, E2);
            put(fri_paths[i].data() + qi * (size_t)fri_depth[i] * 4, (size_t)fri_depth[i] * 4);
        }
    }
    proof->stage_ms[6] = timer.lap();
    {   // the proof has drained (its bytes are on the host): the collectives' event pairs can be read
        float ms = ctx->comm_host_ms;
        const unsigned n_stream = ctx->comm_n & 0xFFFFu;
        for (unsigned i = 0; i < n_stream; i++) {
            float e = 0;
            if (hipEventElapsedTime(&e, ctx->comm_ev[i][0], ctx->comm_ev[i][1]) == hipSuccess) ms += e;
        }
        proof->comm_ms = ms;
        proof->comm_calls = n_stream + (ctx->comm_n >> 16);
        proof->comm_bytes = ctx->comm_bytes;
        for (unsigned i = 0; i < ctx->probe_n; i++) {
            float e = 0;
            if (!ctx->probes[i].closed) continue;   // its closing record failed: the event still belongs to an earlier proof
            if (hipEventElapsedTime(&e, ctx->probes[i].ev[0], ctx->probes[i].ev[1]) != hipSuccess) continue;
            proof->kernel_stats[proof->n_kernel_stats++] = {ctx->probes[i].name, e, ctx->probes[i].bytes};
        }
    }
    proof->ws_high_water = ctx->arena_high_water * 8;
    proof->ws_overflow_slabs = ctx->arena_slabs.size();
    if (ctx->arena_high_water + ((size_t)1 << 17) > ctx->arena_learned) ctx->arena_learned = ctx->arena_high_water + ((size_t)1 << 17);
    guard.ok = true;
    *out = proof;
    return BJ_OK;
}

}  // namespace

extern "C" {

static int stage_witness(bj_ctx *ctx, const bj_setup *S) {
    const size_t n = (size_t)1 << S->log_n, need = (size_t)(S->V + S->Wc + 1) * n;
    if (ctx->wit_stage_elems < need) {   // device staging of the witness, kept for the next proof (no 3 GB hipMalloc per proof)
        if (ctx->wit_stage) BJ_HIP(ctx, hipFree(ctx->wit_stage));
        ctx->wit_stage = nullptr;
        ctx->wit_stage_elems = 0;
        BJ_HIP(ctx, hipMalloc((void **)&ctx->wit_stage, need * 8));
        ctx->wit_stage_elems = need;
    }
    return BJ_OK;
}

int bj_prove(bj_ctx *ctx, const bj_setup *S, const uint64_t *h_variables, const uint64_t *h_multiplicities,
             const uint64_t *h_public_values, bj_proof **out) {
    if (int rc = bj::bind(ctx)) return rc;
    if (!S || !h_variables || !out) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_prove: null argument");
    if (S->lookup_reps && !h_multiplicities) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_prove: multiplicities required");
    if (int rc = stage_witness(ctx, S)) return rc;
    const size_t n = (size_t)1 << S->log_n;
    const unsigned group = bj::env().prove_h2d_group;
    const HostWitness hw{h_variables, h_multiplicities, group, false};
    // the copies are queued inside the proof (after the workspace is reserved); a previous proof on this context has drained
    return prove_impl(ctx, S, ctx->wit_stage, ctx->wit_stage + (size_t)(S->V + S->Wc) * n, h_public_values, out, &hw);
}

}  // extern "C"

namespace bj {
// bj_prove for a lane of bj_prove_async whose sibling is busy: the whole witness crosses PCIe first (one copy on the proof's
// stream, while the other lane's proof has the CUs), then the proof runs as on a resident witness — one leaf kernel instead of
// the group-wise absorption that bj_prove uses to hide the transfer behind its own hashing.  Same bytes either way.
int prove_host_copy_first(bj_ctx *ctx, const bj_setup *S, const uint64_t *h_variables, const uint64_t *h_multiplicities,
                          const uint64_t *h_public_values, bj_proof **out, int mode) {
    if (int rc = bind(ctx)) return rc;
    if (!S || !h_variables || !out) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_prove: null argument");
    if (S->lookup_reps && !h_multiplicities) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_prove: multiplicities required");
    if (int rc = stage_witness(ctx, S)) return rc;
    const size_t n = (size_t)1 << S->log_n, vw = (size_t)(S->V + S->Wc) * n;
    if (mode == 2) {   // transfer and transform in groups as bj_prove does, but ONE leaf kernel at the end: the sibling lane's kernels
                       // cover the transfer, so nothing is gained by absorbing group by group (extra launches, capacity round trips)
        const HostWitness hw{h_variables, h_multiplicities, env().prove_h2d_group, true};
        return prove_impl(ctx, S, ctx->wit_stage, ctx->wit_stage + vw, h_public_values, out, &hw);
    }
    BJ_HIP(ctx, hipMemcpyAsync(ctx->wit_stage, h_variables, vw * 8, hipMemcpyHostToDevice, ctx->stream));
    if (S->lookup_reps) BJ_HIP(ctx, hipMemcpyAsync(ctx->wit_stage + vw, h_multiplicities, n * 8, hipMemcpyHostToDevice, ctx->stream));
    const int rc = prove_impl(ctx, S, ctx->wit_stage, ctx->wit_stage + vw, h_public_values, out, nullptr);
    (void)hipStreamSynchronize(ctx->stream);   // no queued copy may read the caller's witness after this returns
    return rc;
}
}  // namespace bj
