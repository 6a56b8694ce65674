// Seam S3: gates given as op lists (include/boojum_hip.h, bj_gate_program) — what the reference's gpu_synthesizer
// captures from any GateConstraintEvaluator (src/gpu_synthesizer/mod.rs:113-133, 354-444) — evaluated by an interpreter
// kernel.  The four gates of the SHA bench have hand-written evaluators in quotient.hip; everything else goes through
// here: lane = LDE point, the program (relations, constants, writes) is wave-uniform and comes through the scalar cache,
// temporaries live in a per-lane array.  Contribution to the quotient:  T += selector * sum_rep sum_t alpha * term.
#include "ctx.h"
#include "gate_program.h"
#include "gate_program_dev.h"
#include <algorithm>
#include <cstdlib>
#include <vector>

using gl::u64;
using gl::u32;

namespace bj {

namespace {
constexpr int MAX_TMP = BJ_GATE_PROGRAM_MAX_TEMPORARIES;

using namespace gpdev;

// SLOTS > 0: the temporaries live in LDS as [slot][lane] (conflict-free, no HBM-backed scratch traffic: a private array
// indexed by a run-time slot number goes to scratch memory, two loads and a store per recorded operation); after slot
// renaming almost every evaluator needs <= 16 slots.  SLOTS == 0: the private array, for the few large programs.
template <int SLOTS>
__global__ void __launch_bounds__(256) gate_program_kernel(ProgArgs a) {
    __shared__ u64 lds_tmp[SLOTS ? SLOTS : 1][256];
    const size_t I = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= a.Q) return;
    u64 priv[SLOTS ? 1 : MAX_TMP];
    struct Tmp {     // tmp[idx] as an lvalue over either storage
        u64 *p;
        unsigned stride;
        __device__ __forceinline__ u64 &operator[](u32 i) const { return p[(size_t)i * stride]; }
    } tmp{SLOTS ? &lds_tmp[0][threadIdx.x] : priv, SLOTS ? 256u : 1u};
    size_t wb = 0;   // witness column offset of the current repetition
    auto fetch = [&](u32 packed, size_t vb, size_t cb) -> u64 {
        const u32 kind = packed >> 28, idx = packed & 0x0FFFFFFFu;
        switch (kind) {
            case BJ_IDX_VARIABLE_POLY: return gl::canon(a.vars[(vb + idx) * a.var_stride + I]);
            case BJ_IDX_WITNESS_POLY: return gl::canon(a.wits[(wb + idx) * a.var_stride + I]);
            case BJ_IDX_CONSTANT_POLY: return gl::canon(a.consts[(cb + idx) * a.const_stride + I]);
            case BJ_IDX_TEMPORARY: return tmp[idx];
            default: return a.values[idx];
        }
    };
    u64 sel = 1;
    for (unsigned b = 0; b < a.path_len; b++) {
        u64 c = gl::canon(a.consts[(size_t)b * a.const_stride + I]);
        sel = gl::mul(sel, a.path[b] ? c : gl::sub(1, c));
    }
    Acc160g s0, s1;
    s0.clear();
    s1.clear();
    for (unsigned r = 0; r < a.reps; r++) {
        const size_t vb = (size_t)r * a.rep_var_stride, cb = (size_t)a.path_len + (size_t)r * a.rep_const_stride;
        wb = (size_t)r * a.rep_wit_stride;
        for (unsigned i = 0; i < a.n_rel; i++) {
            const DevRelation R = a.rel[i];
            const u64 x = fetch(R.a, vb, cb);
            u64 res;
            switch (R.op) {
                case BJ_OP_ADD: res = gl::add(x, fetch(R.b, vb, cb)); break;
                case BJ_OP_DOUBLE: res = gl::add(x, x); break;
                case BJ_OP_SUB: res = gl::sub(x, fetch(R.b, vb, cb)); break;
                case BJ_OP_NEGATE: res = gl::neg(x); break;
                case BJ_OP_MUL: res = gl::mul(x, fetch(R.b, vb, cb)); break;
                case BJ_OP_SQUARE: res = gl::sqr(x); break;
                default: res = inv_pow(x); break;
            }
            tmp[R.dst] = res;
        }
        for (unsigned t = 0; t < a.n_writes; t++) {
            const u64 term = fetch(a.writes[t], vb, cb);
            if (a.terms) a.terms[((size_t)r * a.n_writes + t) * a.Q + I] = term;
            if (a.alphas) {
                const size_t k = (size_t)r * a.n_writes + t;
                s0.fma(term, a.alphas[2 * k]);
                s1.fma(term, a.alphas[2 * k + 1]);
            }
        }
    }
    if (a.alphas) {
        a.out0[I] = gl::add(gl::canon(a.out0[I]), gl::mul(s0.reduce(), sel));
        a.out1[I] = gl::add(gl::canon(a.out1[I]), gl::mul(s1.reduce(), sel));
    }
}
}  // namespace

// The program as a stream of 32-bit words, section lengths first (era_boojum_amd/gate_codegen.py walks the same way).  Two
// independent 64-bit fingerprints of that stream select a generated kernel: FNV-1a is the key of the switch, the second one is
// compared on a hit, so a program that merely collides with a known one under FNV-1a still runs in the interpreter.
template <typename F>
static void walk_program(const bj_gate_program *p, F mix) {
    mix(p->num_relations);
    mix(p->num_writes);
    for (uint32_t i = 0; i < p->num_relations; i++) {
        const bj_gate_relation &R = p->relations[i];
        const bool binary = R.op == BJ_OP_ADD || R.op == BJ_OP_SUB || R.op == BJ_OP_MUL;
        mix(R.op); mix(R.dst); mix(R.a.kind); mix(R.a.index);
        mix(binary ? R.b.kind : 0u); mix(binary ? R.b.index : 0u);
    }
    mix(p->num_values);
    for (uint32_t i = 0; i < p->num_values; i++) {
        const u64 v = gl::canon(p->values[i]);
        mix((uint32_t)v); mix((uint32_t)(v >> 32));
    }
    for (uint32_t t = 0; t < p->num_writes; t++) { mix(p->writes[t].kind); mix(p->writes[t].index); }
    mix(p->num_temporaries);
}
uint64_t gate_program_hash(const bj_gate_program *p) {
    uint64_t h = 0xcbf29ce484222325ULL;
    walk_program(p, [&](uint32_t w) {
        for (int b = 0; b < 4; b++) {
            h ^= (w >> (8 * b)) & 0xFFu;
            h *= 0x100000001b3ULL;
        }
    });
    return h;
}
uint64_t gate_program_check(const bj_gate_program *p) {
    uint64_t h = 0x9E3779B97F4A7C15ULL;
    walk_program(p, [&](uint32_t w) {
        h ^= w;
        h *= 0xFF51AFD7ED558CCDULL;
        h ^= h >> 32;
    });
    return h;
}

void gate_program_extent(const bj_gate_program *p, unsigned *var_extent, unsigned *const_extent, unsigned *wit_extent) {
    unsigned v = 0, c = 0, w = 0;
    auto see = [&](const bj_gate_index &ix) {
        if (ix.kind == BJ_IDX_VARIABLE_POLY && ix.index + 1 > v) v = ix.index + 1;
        if (ix.kind == BJ_IDX_CONSTANT_POLY && ix.index + 1 > c) c = ix.index + 1;
        if (ix.kind == BJ_IDX_WITNESS_POLY && ix.index + 1 > w) w = ix.index + 1;
    };
    for (uint32_t i = 0; p && p->relations && i < p->num_relations; i++) {
        const bj_gate_relation &R = p->relations[i];
        see(R.a);
        if (R.op == BJ_OP_ADD || R.op == BJ_OP_SUB || R.op == BJ_OP_MUL) see(R.b);
    }
    for (uint32_t t = 0; p && p->writes && t < p->num_writes; t++) see(p->writes[t]);
    *var_extent = v;
    *const_extent = c;
    if (wit_extent) *wit_extent = w;
}

int DevProgram::upload(bj_ctx *ctx, const bj_gate_program *p) {
    if (!p || !p->relations || !p->writes || p->num_writes == 0)
        return fail(ctx, BJ_ERR_INVALID_ARG, "gate program: null / empty program");
    if (p->num_temporaries > (unsigned)MAX_TMP)
        return fail(ctx, BJ_ERR_UNSUPPORTED, "gate program: %u temporaries (at most %d)", p->num_temporaries, MAX_TMP);
    auto check = [&](const bj_gate_index &ix) -> bool {
        switch (ix.kind) {
            case BJ_IDX_VARIABLE_POLY:
            case BJ_IDX_WITNESS_POLY:
            case BJ_IDX_CONSTANT_POLY: return ix.index < (1u << 20);
            case BJ_IDX_TEMPORARY: return ix.index < p->num_temporaries;
            case BJ_IDX_CONSTANT_VALUE: return ix.index < p->num_values && p->values;
            default: return false;
        }
    };
    std::vector<DevRelation> rel(p->num_relations);
    for (uint32_t i = 0; i < p->num_relations; i++) {
        const bj_gate_relation &R = p->relations[i];
        if (R.op < BJ_OP_ADD || R.op > BJ_OP_INVERSE || R.dst >= p->num_temporaries || !check(R.a))
            return fail(ctx, BJ_ERR_INVALID_ARG, "gate program: bad relation %u", i);
        const bool binary = R.op == BJ_OP_ADD || R.op == BJ_OP_SUB || R.op == BJ_OP_MUL;
        if (binary && !check(R.b)) return fail(ctx, BJ_ERR_INVALID_ARG, "gate program: bad second operand in relation %u", i);
        rel[i] = DevRelation{R.op, R.dst, (R.a.kind << 28) | R.a.index, binary ? (R.b.kind << 28) | R.b.index : 0u};
    }
    std::vector<u32> wr(p->num_writes);
    for (uint32_t t = 0; t < p->num_writes; t++) {
        if (!check(p->writes[t])) return fail(ctx, BJ_ERR_INVALID_ARG, "gate program: bad write %u", t);
        wr[t] = (p->writes[t].kind << 28) | p->writes[t].index;
    }
    std::vector<u64> vals(p->num_values ? p->num_values : 1, 0);
    for (uint32_t i = 0; i < p->num_values; i++) vals[i] = gl::canon(p->values[i]);
    n_rel = p->num_relations;
    n_tmp = p->num_temporaries;
    {
        unsigned ve = 0, ce = 0, we = 0;
        gate_program_extent(p, &ve, &ce, &we);
        reads_witness = we != 0;
    }
    hash = gate_program_hash(p);
    this->check = gate_program_check(p);
    n_writes = p->num_writes;
    const size_t bytes = rel.size() * sizeof(DevRelation) + vals.size() * 8 + wr.size() * 4 + 64;
    if (hipMalloc(&block, bytes) != hipSuccess) return fail(ctx, BJ_ERR_OOM, "gate program: allocation failed");
    char *base = (char *)block;
    d_values = (u64 *)base;
    d_rel = (DevRelation *)(base + vals.size() * 8);
    d_writes = (u32 *)(base + vals.size() * 8 + (rel.size() ? rel.size() : 1) * sizeof(DevRelation));
    int rc = bj_memcpy_h2d(ctx, d_values, vals.data(), vals.size() * 8);
    if (!rc && !rel.empty()) rc = bj_memcpy_h2d(ctx, d_rel, rel.data(), rel.size() * sizeof(DevRelation));
    if (!rc) rc = bj_memcpy_h2d(ctx, d_writes, wr.data(), wr.size() * 4);
    return rc;
}
void DevProgram::release() {
    if (block) (void)hipFree(block);
    block = nullptr;
}

static ProgArgs make_prog_args(const DevProgram &P, const u64 *d_vars, size_t var_stride, const u64 *d_consts, size_t const_stride,
                               unsigned path_len, const unsigned char *path, unsigned reps, unsigned rep_var_stride,
                               unsigned rep_const_stride, const u64 *d_alphas, size_t Q, u64 *d_out0, u64 *d_out1, u64 *d_terms,
                               const u64 *d_wits, unsigned rep_wit_stride) {
    ProgArgs a{};
    a.wits = d_wits; a.rep_wit_stride = rep_wit_stride;
    a.vars = d_vars; a.var_stride = var_stride; a.consts = d_consts; a.const_stride = const_stride;
    a.rel = P.d_rel; a.values = P.d_values; a.writes = P.d_writes; a.n_rel = P.n_rel; a.n_writes = P.n_writes;
    a.path_len = path_len;
    for (unsigned b = 0; b < 8; b++) a.path[b] = b < path_len ? path[b] : 0;
    a.reps = reps; a.rep_var_stride = rep_var_stride; a.rep_const_stride = rep_const_stride;
    a.alphas = d_alphas; a.Q = Q; a.out0 = d_out0; a.out1 = d_out1; a.terms = d_terms;
    return a;
}

// the op-list gates of one circuit, quotient mode: the ones with a generated body go out as ONE fused launch when there are
// at least two of them (they all sweep the same general-purpose columns), everything else one launch per gate as before
void launch_gate_programs(const GateLaunch *gates, unsigned n, const u64 *d_vars, size_t var_stride, const u64 *d_consts,
                          size_t const_stride, size_t Q, u64 *d_out0, u64 *d_out1, hipStream_t s, const u64 *d_wits) {
    if (!Q || !n) return;
    static const bool no_aot = getenv("BJ_GATE_NO_AOT") != nullptr, no_fuse = getenv("BJ_GATE_NO_FUSE") != nullptr;
    std::vector<unsigned> fused;
    if (!no_aot && !no_fuse)
        for (unsigned i = 0; i < n; i++) {
            const GateLaunch &G = gates[i];
            if (gate_aot_known(G.program->hash, G.program->check) && !G.program->reads_witness && G.rep_var_stride && G.reps &&
                fused.size() < (size_t)gpdev::BJ_FUSED_MAX)
                fused.push_back(i);
        }
    bool done_fused = false;
    if (fused.size() >= 2) {
        std::vector<ProgArgs> args;
        std::vector<uint64_t> hs, cs;
        for (unsigned i : fused) {
            const GateLaunch &G = gates[i];
            args.push_back(make_prog_args(*G.program, d_vars, var_stride, d_consts, const_stride, G.path_len, G.path, G.reps,
                                          G.rep_var_stride, G.rep_const_stride, G.d_alphas, Q, d_out0, d_out1, nullptr, nullptr, 0));
            hs.push_back(G.program->hash);
            cs.push_back(G.program->check);
        }
        done_fused = launch_gate_aot_fused(hs.data(), cs.data(), args.data(), (unsigned)args.size(), (unsigned)((Q + 255) / 256), s);
    }
    for (unsigned i = 0; i < n; i++) {
        if (done_fused && std::find(fused.begin(), fused.end(), i) != fused.end()) continue;
        const GateLaunch &G = gates[i];
        launch_gate_program(*G.program, d_vars, var_stride, d_consts, const_stride, G.path_len, G.path, G.reps, G.rep_var_stride,
                            G.rep_const_stride, G.d_alphas, Q, d_out0, d_out1, nullptr, s, d_wits, G.rep_wit_stride);
    }
}

void launch_gate_program(const DevProgram &P, const u64 *d_vars, size_t var_stride, const u64 *d_consts, size_t const_stride,
                         unsigned path_len, const unsigned char *path, unsigned reps, unsigned rep_var_stride,
                         unsigned rep_const_stride, const u64 *d_alphas, size_t Q, u64 *d_out0, u64 *d_out1, u64 *d_terms,
                         hipStream_t s, const u64 *d_wits, unsigned rep_wit_stride) {
    const ProgArgs a = make_prog_args(P, d_vars, var_stride, d_consts, const_stride, path_len, path, reps, rep_var_stride,
                                      rep_const_stride, d_alphas, Q, d_out0, d_out1, d_terms, d_wits, rep_wit_stride);
    if (!Q) return;
    const dim3 grid((unsigned)((Q + 255) / 256)), block(256);
    static const bool no_aot = getenv("BJ_GATE_NO_AOT") != nullptr;
    if (!no_aot && launch_gate_aot(P.hash, P.check, a, grid.x, s)) return;   // a generated straight-line kernel exists for this program
    if (P.n_tmp <= 8)
        hipLaunchKernelGGL(gate_program_kernel<8>, grid, block, 0, s, a);
    else if (P.n_tmp <= 16)
        hipLaunchKernelGGL(gate_program_kernel<16>, grid, block, 0, s, a);
    else if (P.n_tmp <= 32)
        hipLaunchKernelGGL(gate_program_kernel<32>, grid, block, 0, s, a);
    else
        hipLaunchKernelGGL(gate_program_kernel<0>, grid, block, 0, s, a);
}

}  // namespace bj

extern "C" int bj_gate_program_generated(const bj_gate_program *program) {
    if (!program || !program->relations || !program->writes) return 0;
    return bj::gate_aot_known(bj::gate_program_hash(program), bj::gate_program_check(program)) ? 1 : 0;
}

extern "C" int bj_gate_program_eval(bj_ctx *ctx, const bj_gate_program *program, const uint64_t *d_vars, size_t var_stride,
                                    const uint64_t *d_consts, size_t const_stride, unsigned num_repetitions,
                                    unsigned rep_var_stride, unsigned rep_const_stride, size_t n_points, uint64_t *d_terms) {
    if (int rc = bj::bind(ctx)) return rc;
    if (!d_vars || !d_terms || num_repetitions == 0) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_gate_program_eval: null argument");
    {
        unsigned ve = 0, ce = 0, we = 0;
        if (program && program->relations && program->writes) bj::gate_program_extent(program, &ve, &ce, &we);
        if (we) return bj::fail(ctx, BJ_ERR_UNSUPPORTED, "bj_gate_program_eval: the stand-alone evaluator takes no witness columns");
    }
    bj::DevProgram P;
    if (int rc = P.upload(ctx, program)) {
        P.release();
        return rc;
    }
    bj::launch_gate_program(P, d_vars, var_stride, d_consts ? d_consts : d_vars, const_stride, 0, nullptr, num_repetitions,
                            rep_var_stride, rep_const_stride, nullptr, n_points, nullptr, nullptr, d_terms, ctx->stream);
    int rc = BJ_OK;
    if (hipGetLastError() != hipSuccess) rc = bj::fail(ctx, BJ_ERR_HIP, "bj_gate_program_eval: launch failed");
    (void)hipStreamSynchronize(ctx->stream);
    P.release();
    return rc;
}
