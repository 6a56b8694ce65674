// Hand-written evaluator of Poseidon2FlattenedGate<8, 12, 4> (src/cs/gates/poseidon2.rs:165-410) for the quotient: the gate
// of the recursion-layer circuits.  One repetition spans 130 variables — 12 inputs, 12 outputs and, from the second full
// round on, a fresh variable for every S-box input ("degree reset") — and pushes 118 terms: state_i - variable at every
// reset, output_i - state_i at the end.  The op-list interpreter (gate_program.hip) runs the same gate as 2.4 k recorded
// operations with 122 temporaries in scratch memory; here the state stays in registers and the terms go straight into the
// alpha-weighted 160-bit accumulators (~10x fewer instructions per LDE point).  Same terms, same order, same proof.
#include "gl.h"
#include "kernels.h"
#include "poseidon_rc.inc"

using gl::u32;
using gl::u64;

namespace bj {
namespace {

__constant__ u64 P2G_RC[BJ_POSEIDON_NUM_RC] = BJ_POSEIDON_RC_TABLE;

struct Acc160p {   // sum of 128-bit products, reduced once
    u32 w[5];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int i = 0; i < 5; i++) w[i] = 0;
    }
    __device__ __forceinline__ void fma(u64 a, u64 b) {
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
    __device__ __forceinline__ u64 reduce() const {
        u64 r = gl::reduce_limbs(w[3], w[2], gl::pack(w[0], w[1]));
        return gl::sub(r, (u64)w[4] << 32);
    }
};

__device__ __forceinline__ u64 pow7(u64 x) {
    u64 x2 = gl::sqr(x), x3 = gl::mul(x2, x), x4 = gl::sqr(x2);
    return gl::mul(x4, x3);
}
__device__ __forceinline__ void m4(u64 *x) {   // [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]] (suggested_mds.rs:21-56)
    u64 t0 = gl::add(x[0], x[1]), t1 = gl::add(x[2], x[3]);
    u64 t2 = gl::add(gl::dbl(x[1]), t1), t3 = gl::add(gl::dbl(x[3]), t0);
    u64 t4 = gl::add(gl::dbl(gl::dbl(t1)), t3), t5 = gl::add(gl::dbl(gl::dbl(t0)), t2);
    x[0] = gl::add(t3, t5); x[1] = t5; x[2] = gl::add(t2, t4); x[3] = t4;
}
__device__ __forceinline__ void ext_mds(u64 (&s)[12]) {
    m4(s); m4(s + 4); m4(s + 8);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        u64 sum = gl::add(gl::add(s[j], s[4 + j]), s[8 + j]);
        s[j] = gl::add(s[j], sum); s[4 + j] = gl::add(s[4 + j], sum); s[8 + j] = gl::add(s[8 + j], sum);
    }
}

__global__ void __launch_bounds__(256)
quotient_poseidon2_flattened_kernel(const u64 *vars, size_t var_stride, const u64 *consts, size_t const_stride,
                                    unsigned path_len, unsigned path_bits, const u64 *alphas /* [118][2] */, size_t Q, u64 *out0,
                                    u64 *out1) {
    const size_t I = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= Q) return;
    u64 sel = 1;
    for (unsigned b = 0; b < path_len; b++) {
        u64 c = gl::canon(consts[(size_t)b * const_stride + I]);
        sel = gl::mul(sel, ((path_bits >> b) & 1u) ? c : gl::sub(1, c));
    }
    Acc160p a0, a1;
    a0.clear();
    a1.clear();
    unsigned term = 0, nxt = 24;
    auto var = [&](unsigned k) { return gl::canon(vars[(size_t)k * var_stride + I]); };
    auto push = [&](u64 t) {
        a0.fma(t, alphas[2 * term]);
        a1.fma(t, alphas[2 * term + 1]);
        term++;
    };
    constexpr unsigned SH[12] = {4, 14, 11, 8, 0, 5, 2, 9, 13, 6, 3, 12};
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = var(i);
    ext_mds(s);
#pragma unroll 1
    for (int rnd = 0; rnd < 4; rnd++) {
        if (rnd) {
#pragma unroll
            for (int i = 0; i < 12; i++) {
                const u64 v = var(nxt++);
                push(gl::sub(s[i], v));
                s[i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = pow7(gl::add(s[i], P2G_RC[12 * rnd + i]));
        ext_mds(s);
    }
#pragma unroll 1
    for (int rnd = 0; rnd < 22; rnd++) {
        s[0] = gl::add(s[0], P2G_RC[12 * (4 + rnd)]);
        const u64 v = var(nxt++);
        push(gl::sub(s[0], v));
        s[0] = pow7(v);
        u64 tot = s[0];
#pragma unroll
        for (int i = 1; i < 12; i++) tot = gl::add(tot, s[i]);
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl::add(gl::mul_pow2(s[i], SH[i]), tot);
    }
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const u64 v = var(nxt++);
            push(gl::sub(s[i], v));
            s[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = pow7(gl::add(s[i], P2G_RC[12 * (26 + k) + i]));
        ext_mds(s);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) push(gl::sub(var(12 + i), s[i]));
    out0[I] = gl::add(gl::canon(out0[I]), gl::mul(a0.reduce(), sel));
    out1[I] = gl::add(gl::canon(out1[I]), gl::mul(a1.reduce(), sel));
}

}  // namespace

void launch_quotient_poseidon2_flattened(const u64 *d_vars, size_t var_stride, const u64 *d_consts, size_t const_stride,
                                         unsigned path_len, const unsigned char *path, const u64 *d_alphas, size_t Q,
                                         u64 *d_out0, u64 *d_out1, hipStream_t s) {
    unsigned bits = 0;
    for (unsigned b = 0; b < path_len; b++) bits |= (path[b] ? 1u : 0u) << b;
    hipLaunchKernelGGL(quotient_poseidon2_flattened_kernel, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, s, d_vars,
                       var_stride, d_consts, const_stride, path_len, bits, d_alphas, Q, d_out0, d_out1);
}

}  // namespace bj
