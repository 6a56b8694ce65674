// Prover round 2 on the device: copy-permutation grand product (z, partial products) and the log-derivative lookup
// polynomials A_i, B, all over the MAIN domain in natural row order (they are interpolated + LDE'd afterwards).
//
// Must equal (as canonical residues):
//   pointwise_rational_in_extension / compute_partial_products_in_extension / shifted_grand_product_in_extension
//                                               src/cs/implementations/copy_permutation.rs:114-248, 649-830, 425-510
//   compute_lookup_poly_pairs_specialized       src/cs/implementations/lookup_argument_in_ext.rs:320-700
// The reference's results do not depend on its thread chunking (products in a commutative ring), so the device may use
// any association order: per-row products, a three-phase device-wide exclusive scan under F_p^2 multiplication.
#include "gl.h"
#include "kernels.h"

using gl::u64;
using gl::u32;

namespace bj {

__device__ inline u64 inv_chain2(u64 x) {   // x^(p-2), p - 2 = (2^32 - 2) * 2^32 + (2^32 - 1)
    auto sqn = [](u64 v, int n) { for (int i = 0; i < n; i++) v = gl::sqr(v); return v; };
    u64 a1 = x, a2 = gl::mul(sqn(a1, 1), a1), a4 = gl::mul(sqn(a2, 2), a2), a8 = gl::mul(sqn(a4, 4), a4);
    u64 a16 = gl::mul(sqn(a8, 8), a8), a24 = gl::mul(sqn(a16, 8), a8), a28 = gl::mul(sqn(a24, 4), a4);
    u64 a30 = gl::mul(sqn(a28, 2), a2), a31 = gl::mul(sqn(a30, 1), a1);
    u64 b = gl::sqr(a31), a32 = gl::mul(b, x);
    return gl::mul(sqn(b, 32), a32);
}
__device__ __forceinline__ gl::e2 e2_inv_dev(gl::e2 a) {
    u64 seven = gl::sub(gl::mul_pow2(gl::sqr(a.c1), 3), gl::sqr(a.c1));
    u64 ni = inv_chain2(gl::sub(gl::sqr(a.c0), seven));
    return {gl::mul(a.c0, ni), gl::neg(gl::mul(a.c1, ni))};
}
// omega_n^r for a natural index r from the bit-reversed forward twiddle table (T[j] = w^bitrev(j), j < n/2)
__device__ __forceinline__ u64 omega_pow_nat(const u64 *tw, unsigned log_n, u32 r) {
    if (log_n == 0) return 1;
    u32 half = 1u << (log_n - 1);
    u64 w = tw[gl::bitrev32(r & (half - 1), log_n - 1)];
    return (r & half) ? gl::neg(w) : w;
}

// P[j][row] = prod_{i in chunk j} (w_i + beta*k_i*x + gamma) / (w_i + beta*sigma_i + gamma);  out: [n_chunks][2][n]
// A lane handles RAT_PTS rows (256 apart, so every access stays coalesced) of one chunk and inverts their denominators
// together (Montgomery's trick: one F_p^2 inversion = one x^(p-2) chain per RAT_PTS rows instead of per row — the inversion
// was 60 % of this kernel's multiplications).
#ifndef BJ_RAT_PTS
#define BJ_RAT_PTS 4
#endif
static constexpr int RAT_PTS = BJ_RAT_PTS;
// SMALLK: every non-residue fits 32 bits (make_non_residues' output always does; the launcher is told): k_c * (x * beta) as a
// 32 x 64-bit product (gl::mul_u32_weak) instead of a 64 x 64 one
template <bool SMALLK>
__global__ void __launch_bounds__(256)
copy_perm_rational_kernel(const u64 *vars, size_t var_stride, const u64 *sigmas, size_t sig_stride, const u64 *non_res,
                          unsigned V, unsigned chunk, unsigned log_n, const u64 *tw, gl::e2 beta, gl::e2 gamma, u64 *out) {
    const size_t n = (size_t)1 << log_n;
    const size_t base = (size_t)blockIdx.x * (256 * RAT_PTS) + threadIdx.x;
    const unsigned j = blockIdx.y;
    gl::e2 num[RAT_PTS], den[RAT_PTS];
    u64 xb0[RAT_PTS], xb1[RAT_PTS];       // x * beta, once per point: a column's k_c * x * beta is then two products instead of three
#pragma unroll
    for (int k = 0; k < RAT_PTS; k++) {
        const size_t r = base + (size_t)k * 256;
        const u64 x = r < n ? omega_pow_nat(tw, log_n, (u32)r) : 0;
        xb0[k] = gl::mul_weak(x, beta.c0);
        xb1[k] = gl::mul_weak(x, beta.c1);
        num[k] = {1, 0};
        den[k] = {1, 0};
    }
    for (unsigned i = j * chunk; i < (j + 1) * chunk && i < V; i++) {
        const u64 kr = SMALLK ? gl::canon(non_res[i]) : non_res[i];
#pragma unroll
        for (int k = 0; k < RAT_PTS; k++) {
            const size_t r = base + (size_t)k * 256;
            if (r >= n) continue;
            // weak residues throughout (any 64-bit representative; see gl::mul_weak): the running products are made canonical
            // once, after the last column, instead of after every operation
            const u64 w = vars[(size_t)i * var_stride + r];
            const u64 wg = gl::add_weak(w, gamma.c0);
            const gl::e2 a = SMALLK ? gl::e2{gl::add_weak(gl::mul_u32_weak(xb0[k], (u32)kr), wg), gl::add_weak(gl::mul_u32_weak(xb1[k], (u32)kr), gamma.c1)}
                                    : gl::e2{gl::add_weak(gl::mul_weak(kr, xb0[k]), wg), gl::add_weak(gl::mul_weak(kr, xb1[k]), gamma.c1)};
            const u64 s = sigmas[(size_t)i * sig_stride + r];
            const gl::e2 b{gl::add_weak(gl::mul_weak(s, beta.c0), wg), gl::add_weak(gl::mul_weak(s, beta.c1), gamma.c1)};
            num[k] = gl::e2_mul_weak(num[k], a);
            den[k] = gl::e2_mul_weak(den[k], b);
        }
    }
#pragma unroll
    for (int k = 0; k < RAT_PTS; k++) {
        num[k] = {gl::canon(num[k].c0), gl::canon(num[k].c1)};
        den[k] = {gl::canon(den[k].c0), gl::canon(den[k].c1)};
    }
    // 1/den[k] for all k from one inversion: pre[k] = den[0..k-1], inv(all) walked back
    gl::e2 pre[RAT_PTS], run{1, 0};
#pragma unroll
    for (int k = 0; k < RAT_PTS; k++) {
        pre[k] = run;
        run = gl::e2_mul(run, den[k]);
    }
    gl::e2 inv_run = e2_inv_dev(run);
#pragma unroll
    for (int k = RAT_PTS - 1; k >= 0; k--) {
        const gl::e2 inv_k = gl::e2_mul(inv_run, pre[k]);
        inv_run = gl::e2_mul(inv_run, den[k]);
        const size_t r = base + (size_t)k * 256;
        if (r < n) {
            gl::e2 p = gl::e2_mul(num[k], inv_k);
            out[((size_t)2 * j) * n + r] = p.c0;
            out[((size_t)2 * j + 1) * n + r] = p.c1;
        }
    }
}

// in place: P[j] <- prod_{j' <= j} P[j'] per row; almost_z = last prefix
__global__ void __launch_bounds__(256) chunk_prefix_kernel(u64 *P, unsigned n_chunks, size_t n) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    gl::e2 acc{P[r], P[n + r]};
    for (unsigned j = 1; j < n_chunks; j++) {
        gl::e2 p{P[((size_t)2 * j) * n + r], P[((size_t)2 * j + 1) * n + r]};
        acc = gl::e2_mul(acc, p);
        P[((size_t)2 * j) * n + r] = acc.c0;
        P[((size_t)2 * j + 1) * n + r] = acc.c1;
    }
}

// ---- exclusive scan under F_p^2 multiplication: z[0] = 1, z[r] = prod_{r' < r} a[r'] ----
static constexpr int SCAN_TPB = 256, SCAN_PER_THREAD = 4, SCAN_BLOCK = SCAN_TPB * SCAN_PER_THREAD;

__device__ __forceinline__ gl::e2 block_exclusive_scan(gl::e2 v, gl::e2 *lds, gl::e2 &total) {
    const unsigned t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (unsigned off = 1; off < SCAN_TPB; off <<= 1) {   // Hillis-Steele inclusive scan
        gl::e2 cur = lds[t];
        gl::e2 other = t >= off ? lds[t - off] : gl::e2{1, 0};
        __syncthreads();
        lds[t] = t >= off ? gl::e2_mul(other, cur) : cur;
        __syncthreads();
    }
    total = lds[SCAN_TPB - 1];
    gl::e2 excl = t ? lds[t - 1] : gl::e2{1, 0};
    __syncthreads();
    return excl;
}

// phase 1: local exclusive scan inside blocks of 1024 rows, block products out
__global__ void __launch_bounds__(SCAN_TPB)
scan_local_kernel(const u64 *a0, const u64 *a1, u64 *z0, u64 *z1, u64 *block_tot, size_t n) {
    __shared__ gl::e2 lds[SCAN_TPB];
    const size_t base = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * SCAN_PER_THREAD;
    gl::e2 v[SCAN_PER_THREAD], pre[SCAN_PER_THREAD];
    gl::e2 run{1, 0};
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) {
        size_t r = base + k;
        v[k] = r < n ? gl::e2{a0[r], a1[r]} : gl::e2{1, 0};
        pre[k] = run;
        run = gl::e2_mul(run, v[k]);
    }
    gl::e2 total;
    gl::e2 excl = block_exclusive_scan(run, lds, total);
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) {
        size_t r = base + k;
        if (r < n) {
            gl::e2 o = gl::e2_mul(excl, pre[k]);
            z0[r] = o.c0;
            z1[r] = o.c1;
        }
    }
    if (threadIdx.x == 0) {
        block_tot[2 * (size_t)blockIdx.x] = total.c0;
        block_tot[2 * (size_t)blockIdx.x + 1] = total.c1;
    }
}
// phase 2: exclusive scan of the block products by ONE workgroup (serial over chunks of 256 with a running prefix)
__global__ void __launch_bounds__(SCAN_TPB) scan_blocks_kernel(u64 *block_tot, size_t n_blocks) {
    __shared__ gl::e2 lds[SCAN_TPB];
    gl::e2 carry{1, 0};
    for (size_t start = 0; start < n_blocks; start += SCAN_TPB) {
        size_t i = start + threadIdx.x;
        gl::e2 v = i < n_blocks ? gl::e2{block_tot[2 * i], block_tot[2 * i + 1]} : gl::e2{1, 0};
        gl::e2 total;
        gl::e2 excl = block_exclusive_scan(v, lds, total);
        if (i < n_blocks) {
            gl::e2 o = gl::e2_mul(carry, excl);
            block_tot[2 * i] = o.c0;
            block_tot[2 * i + 1] = o.c1;
        }
        carry = gl::e2_mul(carry, total);
        __syncthreads();
    }
}
// phase 3: z[r] *= prefix of its block; also emits partial products partial_k = z * Q_k into the stage-2 columns
__global__ void __launch_bounds__(256)
scan_apply_partials_kernel(u64 *z0, u64 *z1, const u64 *block_pre, const u64 *Q, unsigned n_chunks, size_t n,
                           u64 *partials /* [(n_chunks-1)][2][n] */) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    size_t b = r / SCAN_BLOCK;
    gl::e2 z = gl::e2_mul(gl::e2{block_pre[2 * b], block_pre[2 * b + 1]}, gl::e2{z0[r], z1[r]});
    z0[r] = z.c0;
    z1[r] = z.c1;
    for (unsigned k = 0; k + 1 < n_chunks; k++) {
        gl::e2 qk{Q[((size_t)2 * k) * n + r], Q[((size_t)2 * k + 1) * n + r]};
        gl::e2 p = gl::e2_mul(z, qk);
        partials[((size_t)2 * k) * n + r] = p.c0;
        partials[((size_t)2 * k + 1) * n + r] = p.c1;
    }
}

// Driver.  d_tmp must hold 2*n_chunks*n + 2*ceil(n/1024) elements.  Outputs: d_z [2][n], d_partials [(n_chunks-1)][2][n].
void launch_copy_perm_stage2(const u64 *d_vars, size_t var_stride, const u64 *d_sigmas, size_t sig_stride,
                             const u64 *d_non_res, unsigned V, unsigned chunk, unsigned log_n, const u64 *d_tw_fwd,
                             const u64 *beta, const u64 *gamma, u64 *d_tmp, u64 *d_z, u64 *d_partials, hipStream_t s,
                             bool small_non_residues) {
    const size_t n = (size_t)1 << log_n;
    const unsigned n_chunks = (V + chunk - 1) / chunk;
    gl::e2 b{gl::canon(beta[0]), gl::canon(beta[1])}, g{gl::canon(gamma[0]), gl::canon(gamma[1])};
    u64 *P = d_tmp;
    u64 *block_tot = d_tmp + (size_t)2 * n_chunks * n;
    const unsigned rb = (unsigned)((n + 255) / 256);
    if (small_non_residues && !env().copy_perm_wide_k)
        hipLaunchKernelGGL(copy_perm_rational_kernel<true>, dim3((unsigned)((n + 256 * RAT_PTS - 1) / (256 * RAT_PTS)), n_chunks), dim3(256), 0, s, d_vars, var_stride,
                           d_sigmas, sig_stride, d_non_res, V, chunk, log_n, d_tw_fwd, b, g, P);
    else
        hipLaunchKernelGGL(copy_perm_rational_kernel<false>, dim3((unsigned)((n + 256 * RAT_PTS - 1) / (256 * RAT_PTS)), n_chunks), dim3(256), 0, s, d_vars, var_stride,
                           d_sigmas, sig_stride, d_non_res, V, chunk, log_n, d_tw_fwd, b, g, P);
    hipLaunchKernelGGL(chunk_prefix_kernel, dim3(rb), dim3(256), 0, s, P, n_chunks, n);
    const u64 *a0 = P + ((size_t)2 * (n_chunks - 1)) * n, *a1 = a0 + n;
    const size_t n_blocks = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    hipLaunchKernelGGL(scan_local_kernel, dim3((unsigned)n_blocks), dim3(SCAN_TPB), 0, s, a0, a1, d_z, d_z + n, block_tot, n);
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(SCAN_TPB), 0, s, block_tot, n_blocks);
    hipLaunchKernelGGL(scan_apply_partials_kernel, dim3(rb), dim3(256), 0, s, d_z, d_z + n, block_tot, P, n_chunks, n,
                       d_partials);
}

// ---- lookup polynomials: A_i = 1/(beta + sum_j gamma^j col_ij + gamma^w tid),  B = mult/(beta + sum_j gamma^j table_j) ----
// tid = the shared table-id constant column, or (table_id == nullptr) the (w+1)-th variable column of sub-argument i
struct LookupArgs {
    gl::e2 beta;
    gl::e2 gpow[9];   // gamma^0 .. gamma^w  (w <= 8)
};
__global__ void __launch_bounds__(256)
lookup_polys_kernel(const u64 *lvars, size_t var_stride, const u64 *table_id, const u64 *tables, size_t tab_stride,
                    const u64 *mult, unsigned reps, unsigned w, size_t n, LookupArgs a, u64 *outA, u64 *outB) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    // table_id == nullptr: UseSpecializedColumnsWithTableIdAsVariable (lookup_argument_in_ext.rs:354-366) — a sub-argument owns
    // w + 1 variable columns, the last of them the table id; no constant column takes part
    const unsigned cps = table_id ? w : w + 1;
    const u64 tid = table_id ? gl::canon(table_id[r]) : 0;
    // the reps + 1 denominators of a row are inverted in groups of up to LK_GROUP with one F_p^2 inversion per group
    constexpr unsigned LK_GROUP = 9;
    for (unsigned g0 = 0; g0 <= reps; g0 += LK_GROUP) {
        const unsigned cnt = (reps + 1 - g0) < LK_GROUP ? (reps + 1 - g0) : LK_GROUP;
        gl::e2 den[LK_GROUP], pre[LK_GROUP], run{1, 0};
#pragma unroll
        for (unsigned k = 0; k < LK_GROUP; k++) {
            if (k >= cnt) break;
            const unsigned i = g0 + k;                       // i == reps -> the table aggregate (B)
            gl::e2 acc = a.beta;
            if (i < reps) {
                for (unsigned j = 0; j < cps; j++) {
                    u64 v = gl::canon(lvars[(size_t)(i * cps + j) * var_stride + r]);
                    acc = gl::e2_add(acc, gl::e2_mul_base(a.gpow[j], v));
                }
                if (table_id) acc = gl::e2_add(acc, gl::e2_mul_base(a.gpow[w], tid));
            } else {
                for (unsigned j = 0; j <= w; j++) {
                    u64 v = gl::canon(tables[(size_t)j * tab_stride + r]);
                    acc = gl::e2_add(acc, gl::e2_mul_base(a.gpow[j], v));
                }
            }
            den[k] = acc;
            pre[k] = run;
            run = gl::e2_mul(run, acc);
        }
        gl::e2 inv_run = e2_inv_dev(run);
#pragma unroll
        for (int k = (int)LK_GROUP - 1; k >= 0; k--) {
            if ((unsigned)k >= cnt) continue;
            const unsigned i = g0 + (unsigned)k;
            const gl::e2 inv = gl::e2_mul(inv_run, pre[k]);
            inv_run = gl::e2_mul(inv_run, den[k]);
            if (i < reps) {
                outA[((size_t)2 * i) * n + r] = inv.c0;
                outA[((size_t)2 * i + 1) * n + r] = inv.c1;
            } else {
                gl::e2 bb = gl::e2_mul_base(inv, gl::canon(mult[r]));
                outB[r] = bb.c0;
                outB[n + r] = bb.c1;
            }
        }
    }
}

void launch_lookup_polys(const u64 *d_lvars, size_t var_stride, const u64 *d_table_id, const u64 *d_tables,
                         size_t tab_stride, const u64 *d_mult, unsigned reps, unsigned w, unsigned log_n,
                         const u64 *beta, const u64 *gamma, u64 *d_A, u64 *d_B, hipStream_t s) {
    const size_t n = (size_t)1 << log_n;
    LookupArgs a;
    a.beta = {gl::canon(beta[0]), gl::canon(beta[1])};
    gl::e2 g{gl::canon(gamma[0]), gl::canon(gamma[1])};
    a.gpow[0] = {1, 0};
    for (unsigned j = 1; j <= w && j < 9; j++) a.gpow[j] = gl::e2_mul(a.gpow[j - 1], g);
    hipLaunchKernelGGL(lookup_polys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_lvars, var_stride,
                       d_table_id, d_tables, tab_stride, d_mult, reps, w, n, a, d_A, d_B);
}

}  // namespace bj
