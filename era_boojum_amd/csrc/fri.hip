// FRI fold over F_p^2 (two base columns c0, c1 in bit-reversed order) for gfx950.
//
// Must equal fold_multiple (src/cs/implementations/fri/mod.rs:362-474) as driven by interpolate_independent_cosets
// (:476-585) / interpolate_flattened_cosets (:587-678):
//     out[i] = (a + b) + alpha * ((a - b) * roots[i] * coset_inv),   a = in[2i], b = in[2i+1]      (no 1/2 factor)
// roots = the inverse bit-reversed twiddle table of the FULL initial LDE domain (prefix reused at every level),
// coset_inv is squared by the caller after every fold and alpha squared between the folds of one schedule step.
// HBM-bound pointwise kernel: 32 B read + 16 B written per output element, one F_p^2 multiplication.
#include "gl.h"
#include "kernels.h"

using gl::u64;

namespace bj {

__global__ void __launch_bounds__(256)
fri_fold_kernel(const u64 *c0, const u64 *c1, u64 *o0, u64 *o1, const u64 *roots, size_t half, u64 coset_inv, u64 ch0,
                u64 ch1) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    const gl::e2 alpha{ch0, ch1};
    for (; i < half; i += stride) {
        ulonglong2 p0 = reinterpret_cast<const ulonglong2 *>(c0)[i];
        ulonglong2 p1 = reinterpret_cast<const ulonglong2 *>(c1)[i];
        u64 a0 = gl::canon(p0.x), b0 = gl::canon(p0.y), a1 = gl::canon(p1.x), b1 = gl::canon(p1.y);
        u64 r = gl::mul(gl::canon(roots[i]), coset_inv);
        gl::e2 diff{gl::mul(gl::sub(a0, b0), r), gl::mul(gl::sub(a1, b1), r)};
        gl::e2 m = gl::e2_mul(diff, alpha);
        o0[i] = gl::add(gl::add(m.c0, a0), b0);
        o1[i] = gl::add(gl::add(m.c1, a1), b1);
    }
}

// K consecutive folds (one schedule step of 2^K) fused in registers: a lane reads 2^K adjacent values of c0 and of
// c1 (64-512 contiguous bytes), folds them pairwise K times with challenges alpha, alpha^2, alpha^4 and coset
// factors kappa, kappa^2, kappa^4, and writes ONE output.  Intermediate arrays never touch HBM.
template <int K>
__global__ void __launch_bounds__(256)
fri_fold_fused_kernel(const u64 *c0, const u64 *c1, u64 *o0, u64 *o1, const u64 *roots, size_t out_len, size_t j0,
                      u64 coset_inv, u64 ch0, u64 ch1) {
    constexpr int E = 1 << K;
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; j < out_len; j += stride) {
        gl::e2 v[E];
#pragma unroll
        for (int i = 0; i < E; i += 2) {
            ulonglong2 p0 = reinterpret_cast<const ulonglong2 *>(c0 + j * E)[i / 2];
            ulonglong2 p1 = reinterpret_cast<const ulonglong2 *>(c1 + j * E)[i / 2];
            v[i] = {gl::canon(p0.x), gl::canon(p1.x)};
            v[i + 1] = {gl::canon(p0.y), gl::canon(p1.y)};
        }
        gl::e2 alpha{ch0, ch1};
        u64 kappa = coset_inv;
#pragma unroll
        for (int f = 0; f < K; f++) {
            const int outs = E >> (f + 1);
#pragma unroll
            for (int m = 0; m < outs; m++) {
                gl::e2 a = v[2 * m], b = v[2 * m + 1];
                u64 r = gl::mul(gl::canon(roots[(j0 + j) * outs + m]), kappa);   // j0: global output index of local 0
                gl::e2 diff = gl::e2_sub(a, b);
                diff = {gl::mul(diff.c0, r), gl::mul(diff.c1, r)};
                gl::e2 t = gl::e2_mul(diff, alpha);
                v[m] = {gl::add(gl::add(t.c0, a.c0), b.c0), gl::add(gl::add(t.c1, a.c1), b.c1)};
            }
            alpha = gl::e2_sqr(alpha);
            kappa = gl::sqr(kappa);
        }
        o0[j] = v[0].c0;
        o1[j] = v[0].c1;
    }
}

// fold by 2^k in one launch (k = 1..3); len = input length
void launch_fri_fold_step(const u64 *d_c0, const u64 *d_c1, size_t len, unsigned k, u64 *d_o0, u64 *d_o1,
                          const u64 *d_roots, u64 coset_inv, u64 ch0, u64 ch1, hipStream_t s, size_t j0) {
    size_t out_len = len >> k;
    if (!out_len) return;
    unsigned tpb = 256;
    size_t blocks = (out_len + tpb - 1) / tpb;
    if (blocks > 16384) blocks = 16384;
    coset_inv = gl::canon(coset_inv); ch0 = gl::canon(ch0); ch1 = gl::canon(ch1);
    if (k == 1)
        hipLaunchKernelGGL(fri_fold_fused_kernel<1>, dim3((unsigned)blocks), dim3(tpb), 0, s, d_c0, d_c1, d_o0, d_o1, d_roots, out_len, j0, coset_inv, ch0, ch1);
    else if (k == 2)
        hipLaunchKernelGGL(fri_fold_fused_kernel<2>, dim3((unsigned)blocks), dim3(tpb), 0, s, d_c0, d_c1, d_o0, d_o1, d_roots, out_len, j0, coset_inv, ch0, ch1);
    else
        hipLaunchKernelGGL(fri_fold_fused_kernel<3>, dim3((unsigned)blocks), dim3(tpb), 0, s, d_c0, d_c1, d_o0, d_o1, d_roots, out_len, j0, coset_inv, ch0, ch1);
}

void launch_fri_fold(const u64 *d_c0, const u64 *d_c1, size_t len, u64 *d_o0, u64 *d_o1, const u64 *d_roots,
                     u64 coset_inv, u64 ch0, u64 ch1, hipStream_t s) {
    size_t half = len / 2;
    if (!half) return;
    unsigned tpb = 256;
    size_t blocks = (half + tpb - 1) / tpb;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(fri_fold_kernel, dim3((unsigned)blocks), dim3(tpb), 0, s, d_c0, d_c1, d_o0, d_o1, d_roots, half,
                       gl::canon(coset_inv), gl::canon(ch0), gl::canon(ch1));
}

}  // namespace bj
