// Seam S3: gates given as op lists (include/boojum_hip.h, bj_gate_program) — what the reference's gpu_synthesizer
// captures from any GateConstraintEvaluator (src/gpu_synthesizer/mod.rs:113-133, 354-444) — evaluated by an interpreter
// kernel.  The four gates of the SHA bench have hand-written evaluators in quotient.hip; everything else goes through
// here: lane = LDE point, the program (relations, constants, writes) is wave-uniform and comes through the scalar cache,
// temporaries live in a per-lane array.  Contribution to the quotient:  T += selector * sum_rep sum_t alpha * term.
#include "ctx.h"
#include "gate_canon.h"
#include "gate_jit.h"
#include "gate_program.h"
#include "gate_program_dev.h"
#include <algorithm>
#include <cstdlib>
#include <vector>

using gl::u64;
using gl::u32;

namespace bj {
void launch_quotient_poseidon2_flattened(const u64 *d_vars, size_t var_stride, const u64 *d_consts, size_t const_stride,
                                         unsigned path_len, const unsigned char *path, const u64 *d_alphas, size_t Q,
                                         u64 *d_out0, u64 *d_out1, hipStream_t s);

namespace {
constexpr int MAX_TMP = BJ_GATE_PROGRAM_MAX_TEMPORARIES;

using namespace gpdev;

// SLOTS > 0: the temporaries live in LDS as [slot][lane] (conflict-free, no HBM-backed scratch traffic: a private array
// indexed by a run-time slot number goes to scratch memory, two loads and a store per recorded operation); in canonical form
// (gate_canon.h) almost every evaluator needs <= 16 slots.  SLOTS == 0: a private array of PRIV slots, for the few large ones.
// The program is the canonical schedule: OP_WRITE pseudo relations hand a term over as soon as its value exists.
template <int SLOTS, int PRIV>
__global__ void __launch_bounds__(256) gate_program_kernel(ProgArgs a) {
    __shared__ u64 lds_tmp[SLOTS ? SLOTS : 1][256];
    const size_t I = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= a.Q) return;
    u64 priv[SLOTS ? 1 : PRIV];
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
                case BJ_OP_INVERSE: res = inv_pow(x); break;
                default: {   // canon::OP_WRITE: x is term R.dst of this repetition
                    const size_t k = (size_t)r * a.n_writes + R.dst;
                    if (a.terms) a.terms[k * a.Q + I] = x;
                    if (a.alphas) {
                        s0.fma(x, a.alphas[2 * k]);
                        s1.fma(x, a.alphas[2 * k + 1]);
                    }
                    continue;
                }
            }
            tmp[R.dst] = res;
        }
    }
    if (a.alphas) {
        a.out0[I] = gl::add(gl::canon(a.out0[I]), gl::mul(s0.reduce(), sel));
        a.out1[I] = gl::add(gl::canon(a.out1[I]), gl::mul(s1.reduce(), sel));
    }
}
}  // namespace

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

// Whatever numbering and order the host's list has (the reference hands over one fresh temporary per operation from a
// process-wide counter, gpu_synthesizer/mod.rs:210-352), the device sees the canonical schedule of gate_canon.h: slots by
// live range, and a fingerprint of the evaluator's function that finds a generated kernel or names the one compiled here.
int DevProgram::upload(bj_ctx *ctx, const bj_gate_program *p) {
    canon::Program C;
    std::string err;
    if (int rc = canon::canonicalize(p, &C, &err)) return fail(ctx, rc, "%s", err.c_str());
    if (C.num_slots > (unsigned)BJ_GATE_PROGRAM_MAX_SLOTS)
        return fail(ctx, BJ_ERR_UNSUPPORTED, "gate program: %u values live at once (at most %d)", C.num_slots, BJ_GATE_PROGRAM_MAX_SLOTS);
    auto pack = [&](const canon::Operand &x) -> uint32_t {
        return (x.kind << 28) | (x.kind == BJ_IDX_TEMPORARY ? C.slot_of[x.index] : x.index);
    };
    std::vector<DevRelation> rel(C.nodes.size());
    for (size_t i = 0; i < C.nodes.size(); i++) {
        const canon::Node &n = C.nodes[i];
        const bool binary = n.op == BJ_OP_ADD || n.op == BJ_OP_SUB || n.op == BJ_OP_MUL;
        rel[i] = DevRelation{n.op, n.dst, pack(n.a), binary ? pack(n.b) : 0u};
    }
    std::vector<u64> vals = C.values;
    if (vals.empty()) vals.push_back(0);
    n_rel = (unsigned)rel.size();
    n_tmp = C.num_slots;
    n_writes = C.num_terms;
    var_extent = C.var_extent; const_extent = C.const_extent; wit_extent = C.wit_extent;
    reads_witness = C.wit_extent != 0;
    fp[0] = C.fp[0];
    fp[1] = C.fp[1];
    // the fingerprint selects build-time kernels and the hand-written Poseidon2 evaluator: a hit whose structural summary is not
    // the recorded one (a collision of the non-cryptographic mix, accidental or constructed) is not a hit — the program keeps a
    // fingerprint no table knows and goes to the run-time compiler / interpreter, which work from the op list itself
    if ((gate_aot_known(fp[0], fp[1]) || gate_is_poseidon2_flattened(fp[0], fp[1])) &&
        !gate_aot_summary_matches(fp[0], C.num_ops, C.num_slots, C.num_terms, C.var_extent, C.const_extent, C.wit_extent))
        fp[0] = fp[1] = 0;
    const size_t bytes = rel.size() * sizeof(DevRelation) + vals.size() * 8 + 64;
    if (hipMalloc(&block, bytes) != hipSuccess) return fail(ctx, BJ_ERR_OOM, "gate program: allocation failed");
    char *base = (char *)block;
    d_values = (u64 *)base;
    d_rel = (DevRelation *)(base + vals.size() * 8);
    int rc = bj_memcpy_h2d(ctx, d_values, vals.data(), vals.size() * 8);
    if (!rc) rc = bj_memcpy_h2d(ctx, d_rel, rel.data(), rel.size() * sizeof(DevRelation));
    // no build-time kernel for this function: compile one now (gate_jit.hip); the interpreter remains the fallback when the
    // run-time compiler is not installed or BJ_GATE_NO_JIT is set
    if (!rc && !gate_aot_known(fp[0], fp[1]) && !gate_is_poseidon2_flattened(fp[0], fp[1])) jit = jit_gate_kernel(ctx, C);
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
    a.rel = P.d_rel; a.values = P.d_values; a.n_rel = P.n_rel; a.n_writes = P.n_writes;
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
    const bool no_aot = bj::env().gate_no_aot, no_fuse = bj::env().gate_no_fuse;
    std::vector<unsigned> fused;
    if (!no_aot && !no_fuse)
        for (unsigned i = 0; i < n; i++) {
            const GateLaunch &G = gates[i];
            if (gate_aot_fusable(G.program->fp[0], G.program->fp[1]) && !G.program->reads_witness && G.rep_var_stride && G.reps &&
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
            hs.push_back(G.program->fp[0]);
            cs.push_back(G.program->fp[1]);
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
    const bool no_aot = bj::env().gate_no_aot;
    if (!no_aot && launch_gate_aot(P.fp[0], P.fp[1], a, grid.x, s)) return;   // a build-time generated straight-line kernel
    // the reference's own capture of the Poseidon2 flattened gate: the hand-written evaluator (quotient mode, one repetition)
    if (!no_aot && d_alphas && !d_terms && reps == 1 && gate_is_poseidon2_flattened(P.fp[0], P.fp[1])) {
        launch_quotient_poseidon2_flattened(d_vars, var_stride, d_consts, const_stride, path_len, path, d_alphas, Q, d_out0, d_out1, s);
        return;
    }
    if (P.jit && launch_jit_gate(P.jit, a, grid.x, s)) return;                // compiled from this very op list at upload
    if (P.n_tmp <= 8)
        hipLaunchKernelGGL((gate_program_kernel<8, 1>), grid, block, 0, s, a);
    else if (P.n_tmp <= 16)
        hipLaunchKernelGGL((gate_program_kernel<16, 1>), grid, block, 0, s, a);
    else if (P.n_tmp <= 32)
        hipLaunchKernelGGL((gate_program_kernel<32, 1>), grid, block, 0, s, a);
    else if (P.n_tmp <= MAX_TMP)
        hipLaunchKernelGGL((gate_program_kernel<0, MAX_TMP>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((gate_program_kernel<0, BJ_GATE_PROGRAM_MAX_SLOTS>), grid, block, 0, s, a);
}

}  // namespace bj

extern "C" int bj_gate_program_generated(const bj_gate_program *program) {
    bj::canon::Program C;
    std::string err;
    if (bj::canon::canonicalize(program, &C, &err)) return 0;
    return ((bj::gate_aot_known(C.fp[0], C.fp[1]) || bj::gate_is_poseidon2_flattened(C.fp[0], C.fp[1])) &&
            bj::gate_aot_summary_matches(C.fp[0], C.num_ops, C.num_slots, C.num_terms, C.var_extent, C.const_extent, C.wit_extent)) ? 1 : 0;
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
