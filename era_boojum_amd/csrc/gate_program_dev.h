// Internal: what the op-list interpreter (gate_program.hip) and the generated straight-line kernels (gate_aot.hip, emitted by
// era_boojum_amd/gate_codegen.py) share — the launch arguments, the lazy alpha accumulator, the in-kernel inversion.
#pragma once
#include "gate_program.h"
#include "gl.h"

namespace bj {
namespace gpdev {
using gl::u32;
using gl::u64;

struct Acc160g {   // same lazy accumulator as quotient.hip
    u32 w[5];
    __host__ __device__ __forceinline__ void clear() {
#pragma unroll
        for (int i = 0; i < 5; i++) w[i] = 0;
    }
    __host__ __device__ __forceinline__ void fma(u64 a, u64 b) {
        u32 hh, hl;
        u64 lo;
        gl::mul_limbs(a, b, hh, hl, lo);
        u32 c;
        w[0] = __builtin_addc(w[0], gl::lo32(lo), 0u, &c);
        w[1] = __builtin_addc(w[1], gl::hi32(lo), c, &c);
        w[2] = __builtin_addc(w[2], hl, c, &c);
        w[3] = __builtin_addc(w[3], hh, c, &c);
        w[4] += c;
    }
    __host__ __device__ __forceinline__ u64 reduce() const {
        u64 r = gl::reduce_limbs(w[3], w[2], gl::pack(w[0], w[1]));
        return gl::sub(r, (u64)w[4] << 32);
    }
};

__host__ __device__ inline u64 inv_pow(u64 x) {   // x^(p-2); inverse of 0 is 0 like the reference's batch inversion never sees
    u64 r = 1, b = x;
    u64 e = gl::P - 2;
    for (int i = 0; i < 64; i++) {
        if ((e >> i) & 1) r = gl::mul(r, b);
        b = gl::sqr(b);
    }
    return r;
}

struct ProgArgs {
    const u64 *vars;
    size_t var_stride;
    const u64 *consts;
    size_t const_stride;
    const DevRelation *rel;
    const u64 *values;
    const u32 *writes;   // kind << 28 | index
    unsigned n_rel, n_writes;
    unsigned path_len;
    unsigned char path[8];
    unsigned reps, rep_var_stride, rep_const_stride;
    const u64 *wits;     // witness (non-copiable) columns, same stride as vars; nullptr when the program reads none
    unsigned rep_wit_stride;
    const u64 *alphas;   // [reps * n_writes][2] for this gate, or nullptr
    size_t Q;
    u64 *out0, *out1;    // accumulated into (quotient mode)
    u64 *terms;          // raw terms (stand-alone mode)
};
// Several generated evaluators in ONE launch (the gates of a circuit that all sweep the same general-purpose columns): the
// gates advance together over windows of `window` columns, so a column is read from HBM by the first gate that needs it and
// from cache by the others.  sum_g sel_g * sum_t alpha_t term_t is accumulated as sum (sel_g term_t) alpha_t in one pair of
// lazy accumulators — the same field element.
constexpr int BJ_FUSED_MAX = 8;
struct FusedArgs {
    ProgArgs g[BJ_FUSED_MAX];
    int id[BJ_FUSED_MAX];    // index of the generated body
    int n;
    unsigned window, span;   // columns per window; columns covered by the widest gate
};
}  // namespace gpdev
// true if every program has a generated body and the launch was made (quotient mode only: alphas, out0 / out1 of args[0])
bool launch_gate_aot_fused(const uint64_t *hashes, const uint64_t *checks, const gpdev::ProgArgs *args, unsigned n, unsigned blocks,
                           hipStream_t s);
// true if a generated kernel exists for the program with this hash (and it was launched)
bool launch_gate_aot(uint64_t hash, uint64_t check, const gpdev::ProgArgs &a, unsigned blocks, hipStream_t s);
bool gate_aot_known(uint64_t hash, uint64_t check);
uint64_t gate_program_hash(const bj_gate_program *p);
uint64_t gate_program_check(const bj_gate_program *p);
}  // namespace bj
