// Internal: what the op-list interpreter (gate_program.hip), the build-time generated straight-line kernels (gate_aot.hip) and
// the kernels compiled at run time from a host's own op lists (gate_jit.hip) share on the device side — launch arguments, the
// lazy alpha accumulator, the in-kernel inversion, the per-point drivers around a straight-line body.  Depends on gl.h only,
// so that hiprtc can compile it from the copy embedded in the library (jit_headers.inc).
#pragma once
#include "gl.h"

namespace bj {
struct DevRelation {
    uint32_t op, dst, a, b;   // a, b: kind << 28 | index; op 8 (canon::OP_WRITE): a is term `dst` of the repetition
};
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

namespace gpaot {
using namespace gpdev;

#define VAR(k) gl::canon(a.vars[(vb + (k)) * a.var_stride + I])
#define CON(k) gl::canon(a.consts[(cb + (k)) * a.const_stride + I])
#define WIT(k) gl::canon(a.wits[((size_t)r * a.rep_wit_stride + (k)) * a.var_stride + I])
#define BJ_AOT_BODY(NAME, NT_, ...)                                                                                     \
    struct Body_##NAME {                                                                                                \
        static constexpr int NT = NT_;                                                                                  \
        static __host__ __device__ __forceinline__ void run(const ProgArgs &a, size_t I, unsigned r, u64 (&term)[NT_]) { \
            const size_t vb = (size_t)r * a.rep_var_stride, cb = (size_t)a.path_len + (size_t)r * a.rep_const_stride;   \
            (void)vb;                                                                                                   \
            (void)cb;                                                                                                   \
            { __VA_ARGS__ }                                                                                             \
        }                                                                                                               \
    };

__host__ __device__ __forceinline__ u64 selector_at(const ProgArgs &a, size_t I) {
    u64 sel = 1;
    for (unsigned b = 0; b < a.path_len; b++) {
        const u64 c = gl::canon(a.consts[(size_t)b * a.const_stride + I]);
        sel = gl::mul(sel, a.path[b] ? c : gl::sub(1, c));
    }
    return sel;
}

// one LDE point of one gate: raw terms (stand-alone mode) and / or out += sel * sum alpha * term (quotient mode)
template <class B>
__host__ __device__ __forceinline__ void aot_point(const ProgArgs &a, size_t I) {
    const u64 sel = selector_at(a, I);
    Acc160g acc0, acc1;
    acc0.clear();
    acc1.clear();
    for (unsigned r = 0; r < a.reps; r++) {
        u64 term[B::NT];
        B::run(a, I, r, term);
        for (int t = 0; t < B::NT; t++) {
            if (a.terms) a.terms[((size_t)r * B::NT + t) * a.Q + I] = term[t];
            if (a.alphas) {
                const size_t k = (size_t)r * B::NT + t;
                acc0.fma(term[t], a.alphas[2 * k]);
                acc1.fma(term[t], a.alphas[2 * k + 1]);
            }
        }
    }
    if (a.alphas) {
        a.out0[I] = gl::add(gl::canon(a.out0[I]), gl::mul(acc0.reduce(), sel));
        a.out1[I] = gl::add(gl::canon(a.out1[I]), gl::mul(acc1.reduce(), sel));
    }
}

// repetitions [r_lo, r_hi) of one gate inside a fused sweep: (sel * term) * alpha into the shared accumulators
template <class B>
__host__ __device__ __forceinline__ void fused_reps(const ProgArgs &a, size_t I, unsigned r_lo, unsigned r_hi, u64 sel, Acc160g &acc0,
                                                    Acc160g &acc1) {
    for (unsigned r = r_lo; r < r_hi; r++) {
        u64 term[B::NT];
        B::run(a, I, r, term);
        for (int t = 0; t < B::NT; t++) {
            const size_t k = (size_t)r * B::NT + t;
            const u64 st = gl::mul(term[t], sel);
            acc0.fma(st, a.alphas[2 * k]);
            acc1.fma(st, a.alphas[2 * k + 1]);
        }
    }
}
}  // namespace gpaot
}  // namespace bj
