// Openings: barycentric evaluation at z and the DEEP quotient accumulation, for gfx950.
//
// Must equal (as canonical residues):
//   precompute_for_barycentric_evaluation_in_extension      src/cs/implementations/utils.rs:907-1021
//   barycentric_evaluate_{base,extension}_at_extension_...  src/cs/implementations/utils.rs:1085-1242
//   quotening_operation_in_extension                        src/cs/implementations/prover.rs:2523-2706
//
// Both are "many columns x many points" contractions with F_p^2 coefficients:
//   barycentric:  out[col] = sum_i  f_col[i] * w[i]                  (reduction over the points of coset 0)
//   DEEP:         dst[I]  += ( sum_k coef_k * f_k[I]  -  C ) / (x_I - at)   (reduction over the columns)
// A column is base-field, so coef * f is two base multiplications; an F_p^2 column (c0, c1) is presented by the host
// as two base columns with coefficients (ch0, ch1) and (7*ch1, ch0) — same residues, no special case on the device.
// The sums are accumulated UNREDUCED: each 64x64 product is added as a 128-bit integer into a 160-bit accumulator
// (five 32-bit words, carry chain) and reduced mod p once at the end, which replaces a modular reduction + modular add
// per term (~28 VALU ops) by the four mads + five add-with-carry (~10 ops).  HBM: every column value is read once.
#include "gl.h"
#include "kernels.h"

using gl::u64;
using gl::u32;

namespace bj {

struct Acc160 {
    u32 w[5];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int i = 0; i < 5; i++) w[i] = 0;
    }
    // += a * b  (any u64 operands)
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
    // canonical residue of the accumulated integer: 2^64 = 2^32-1, 2^96 = -1, 2^128 = -2^32 (mod p)
    __device__ __forceinline__ u64 reduce() const {
        u64 r = gl::reduce_limbs(w[3], w[2], gl::pack(w[0], w[1]));
        return gl::sub(r, (u64)w[4] << 32);  // w[4] < 2^31 terms, so (w4 << 32) < p
    }
};

// x^(p-2) with an addition chain: p - 2 = (2^32 - 2) * 2^32 + (2^32 - 1)   (~72 multiplications)
__device__ inline u64 inv_chain(u64 x) {
    auto sqn = [](u64 v, int n) {
        for (int i = 0; i < n; i++) v = gl::sqr(v);
        return v;
    };
    u64 a1 = x;                          // 2^1 - 1
    u64 a2 = gl::mul(sqn(a1, 1), a1);    // 2^2 - 1
    u64 a4 = gl::mul(sqn(a2, 2), a2);
    u64 a8 = gl::mul(sqn(a4, 4), a4);
    u64 a16 = gl::mul(sqn(a8, 8), a8);
    u64 a24 = gl::mul(sqn(a16, 8), a8);
    u64 a28 = gl::mul(sqn(a24, 4), a4);
    u64 a30 = gl::mul(sqn(a28, 2), a2);
    u64 a31 = gl::mul(sqn(a30, 1), a1);  // 2^31 - 1
    u64 b = gl::sqr(a31);                // 2^32 - 2
    u64 a32 = gl::mul(b, x);             // 2^32 - 1
    return gl::mul(sqn(b, 32), a32);
}

__device__ __forceinline__ u64 mul7(u64 a) { return gl::sub(gl::mul_pow2(a, 3), a); }

// ---------------------------------------------------------------------------------------------------------
// barycentric weights, stored bit-reversed:  w[j] = cf * omega^i / (z - coset*omega^i),  i = bitrev(j)
// omega^i = T[j>>1] * (-1)^(j&1) with T the bit-reversed forward twiddle table of size n.
// ---------------------------------------------------------------------------------------------------------
__global__ void barycentric_weights_kernel(u64 *w0, u64 *w1, const u64 *tw, unsigned log_n, u64 coset, u64 z0, u64 z1,
                                           u64 cf0, u64 cf1) {
    size_t n = (size_t)1 << log_n;
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    u64 wi = tw[j >> 1];
    if (j & 1) wi = gl::neg(wi);
    u64 x = gl::mul(coset, wi);
    // 1 / (z - x) in F_p^2
    u64 d0 = gl::sub(z0, x), d1 = z1;
    u64 norm = gl::sub(gl::sqr(d0), mul7(gl::sqr(d1)));
    u64 ni = inv_chain(norm);
    gl::e2 inv{gl::mul(d0, ni), gl::neg(gl::mul(d1, ni))};
    gl::e2 c{gl::mul(cf0, wi), gl::mul(cf1, wi)};
    gl::e2 r = gl::e2_mul(inv, c);
    w0[j] = r.c0;
    w1[j] = r.c1;
}

void launch_barycentric_weights(u64 *d_w0, u64 *d_w1, const u64 *d_tw_fwd, unsigned log_n, u64 coset, const u64 *at,
                                hipStream_t s) {
    size_t n = (size_t)1 << log_n;
    if (n == 1) {  // utils.rs:919-927
        u64 one = 1, zero = 0;
        (void)hipMemcpyAsync(d_w0, &one, 8, hipMemcpyHostToDevice, s);
        (void)hipMemcpyAsync(d_w1, &zero, 8, hipMemcpyHostToDevice, s);
        (void)hipStreamSynchronize(s);
        return;
    }
    coset = gl::canon(coset);
    gl::e2 z{gl::canon(at[0]), gl::canon(at[1])};
    // constant factor = coset * (z^n - coset^n) / (n * coset^n)
    u64 t = gl::pow(coset, n);
    gl::e2 zn{1, 0}, base = z;
    for (size_t e = n; e; e >>= 1) {
        if (e & 1) zn = gl::e2_mul(zn, base);
        base = gl::e2_sqr(base);
    }
    zn.c0 = gl::sub(zn.c0, t);
    gl::e2 cf = gl::e2_mul_base(zn, coset);
    cf = gl::e2_mul_base(cf, gl::inv(gl::mul(t, gl::canon((u64)n % gl::P))));
    unsigned tpb = 256;
    hipLaunchKernelGGL(barycentric_weights_kernel, dim3((unsigned)((n + tpb - 1) / tpb)), dim3(tpb), 0, s, d_w0, d_w1,
                       d_tw_fwd, log_n, coset, z.c0, z.c1, cf.c0, cf.c1);
}

// ---------------------------------------------------------------------------------------------------------
// barycentric evaluation of a batch of base columns:  partial[col][block] = sum over the block's points of f * w
// ---------------------------------------------------------------------------------------------------------
static constexpr int BARY_PTS = 32;   // points per lane: a workgroup covers 256 * BARY_PTS consecutive points
static constexpr int BARY_STEP = 4;   // points in flight per column
static constexpr int BARY_COLS = 8;   // columns sharing one read of the weights

// A lane walks its BARY_PTS points four at a time and keeps one 160-bit accumulator pair per column for the whole walk: the
// weights are read once per eight columns, and the cross-lane reduction — the larger half of this kernel when it ran after
// every eight points of every column — happens once per workgroup, for all sixteen sums together, through a transposed LDS tree.
__global__ void __launch_bounds__(256)
barycentric_partial_kernel(const u64 *const *cols, unsigned n_cols, size_t n, const u64 *w0, const u64 *w1,
                           u64 *partials, unsigned n_blocks) {
    __shared__ u64 red[2 * BARY_COLS][256 + 1];
    const unsigned t = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * (256 * BARY_PTS) + t;
    const unsigned c0 = blockIdx.y * BARY_COLS;
    const u64 *f[BARY_COLS];
#pragma unroll
    for (int c = 0; c < BARY_COLS; c++) f[c] = cols[c0 + c < n_cols ? c0 + c : c0];   // a short last group re-reads its first column; the surplus sums are dropped
    Acc160 s0[BARY_COLS], s1[BARY_COLS];
#pragma unroll
    for (int c = 0; c < BARY_COLS; c++) {
        s0[c].clear();
        s1[c].clear();
    }
#pragma unroll 1
    for (int k0 = 0; k0 < BARY_PTS; k0 += BARY_STEP) {
        u64 a[BARY_STEP], b[BARY_STEP];
#pragma unroll
        for (int k = 0; k < BARY_STEP; k++) {   // past the end: a zero weight on a clamped (valid) index — no divergent loads
            const size_t i = base + (size_t)(k0 + k) * 256;
            const size_t ic = i < n ? i : n - 1;
            const u64 wa = w0[ic], wb = w1[ic];
            a[k] = i < n ? wa : 0;
            b[k] = i < n ? wb : 0;
        }
#pragma unroll
        for (int c = 0; c < BARY_COLS; c++) {
            u64 v[BARY_STEP];
#pragma unroll
            for (int k = 0; k < BARY_STEP; k++) {
                const size_t i = base + (size_t)(k0 + k) * 256;
                v[k] = f[c][i < n ? i : n - 1];
            }
#pragma unroll
            for (int k = 0; k < BARY_STEP; k++) {
                s0[c].fma(v[k], a[k]);
                s1[c].fma(v[k], b[k]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < BARY_COLS; c++) {
        red[2 * c][t] = s0[c].reduce();
        red[2 * c + 1][t] = s1[c].reduce();
    }
    __syncthreads();
    {   // lane t: sum j = t & 15, the sixteen lanes' values of segment t >> 4; then sixteen lanes add the segments
        const unsigned j = t & 15, seg = t >> 4;
        u64 acc = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) acc = gl::add(acc, red[j][seg * 16 + k]);
        __syncthreads();
        red[j][seg] = acc;
        __syncthreads();
        if (t < 2 * BARY_COLS && c0 + (t >> 1) < n_cols) {
            u64 tot = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) tot = gl::add(tot, red[t][k]);
            partials[((size_t)(c0 + (t >> 1)) * n_blocks + blockIdx.x) * 2 + (t & 1)] = tot;
        }
    }
}

__global__ void barycentric_final_kernel(const u64 *partials, unsigned n_blocks, u64 *out) {
    __shared__ u64 red[2][256];
    const unsigned t = threadIdx.x, c = blockIdx.x;
    u64 s0 = 0, s1 = 0;
    for (unsigned b = t; b < n_blocks; b += 256) {
        s0 = gl::add(s0, partials[((size_t)c * n_blocks + b) * 2 + 0]);
        s1 = gl::add(s1, partials[((size_t)c * n_blocks + b) * 2 + 1]);
    }
    red[0][t] = s0;
    red[1][t] = s1;
    __syncthreads();
    for (unsigned stride = 128; stride > 0; stride >>= 1) {
        if (t < stride) {
            red[0][t] = gl::add(red[0][t], red[0][t + stride]);
            red[1][t] = gl::add(red[1][t], red[1][t + stride]);
        }
        __syncthreads();
    }
    if (t == 0) {
        out[2 * c] = red[0][0];
        out[2 * c + 1] = red[1][0];
    }
}

unsigned barycentric_num_blocks(size_t n) { return (unsigned)((n + 256 * BARY_PTS - 1) / (256 * BARY_PTS)); }

void launch_barycentric_eval(const u64 *const *d_col_ptrs, unsigned n_cols, size_t n, const u64 *d_w0, const u64 *d_w1,
                             u64 *d_partials, u64 *d_out, hipStream_t s) {
    unsigned nb = barycentric_num_blocks(n);
    hipLaunchKernelGGL(barycentric_partial_kernel, dim3(nb, (n_cols + BARY_COLS - 1) / BARY_COLS), dim3(256), 0, s,
                       d_col_ptrs, n_cols, n, d_w0, d_w1, d_partials, nb);
    hipLaunchKernelGGL(barycentric_final_kernel, dim3(n_cols), dim3(256), 0, s, d_partials, nb, d_out);
}

// ---------------------------------------------------------------------------------------------------------
// out[i] = sum_k (coef0_k + coef1_k u) * col_k[i]  over n entries — the DEEP numerator of a large opening set taken on the
// MONOMIAL forms: sum_k c_k f_k is a polynomial of degree < n, so it is combined once over n coefficients (1/lde_factor of
// the LDE-domain traffic) and extended to the FRI domain by one two-column LDE; exact arithmetic, identical values.
// ---------------------------------------------------------------------------------------------------------
// PTS points per lane, CU columns in flight.  With few points (a 2^14-row proof has 64 wavefronts of them) the kernel is one
// long chain of dependent loads — column pointer, then value — per column: the small-n variant keeps 8 columns in flight
// and one point per lane (234 -> ~40 us at 2^14); the large-n one is throughput-bound and keeps 4 points per lane.
template <int PTS, int CU>
__global__ void __launch_bounds__(256)
linear_combination_kernel(const u64 *const *cols, const u64 *coefs /*[n_cols][2]*/, unsigned n_cols, size_t n, u64 *out0,
                          u64 *out1) {
    const size_t base = (size_t)blockIdx.x * (256 * PTS) + threadIdx.x;
    Acc160 s0[PTS], s1[PTS];
#pragma unroll
    for (int k = 0; k < PTS; k++) {
        s0[k].clear();
        s1[k].clear();
    }
    unsigned c = 0;
    for (; c + CU <= n_cols; c += CU) {
        const u64 *f[CU];
        u64 v[CU][PTS];
#pragma unroll
        for (int j = 0; j < CU; j++) f[j] = cols[c + j];
#pragma unroll
        for (int j = 0; j < CU; j++)
#pragma unroll
            for (int k = 0; k < PTS; k++) {
                const size_t i = base + (size_t)k * 256;
                v[j][k] = i < n ? f[j][i] : 0;
            }
#pragma unroll
        for (int j = 0; j < CU; j++) {
            const u64 a = coefs[2 * (c + j)], b = coefs[2 * (c + j) + 1];
#pragma unroll
            for (int k = 0; k < PTS; k++) {
                s0[k].fma(v[j][k], a);
                s1[k].fma(v[j][k], b);
            }
        }
    }
    for (; c < n_cols; c++) {
        const u64 *f = cols[c];
        const u64 a = coefs[2 * c], b = coefs[2 * c + 1];
#pragma unroll
        for (int k = 0; k < PTS; k++) {
            const size_t i = base + (size_t)k * 256;
            const u64 v = i < n ? f[i] : 0;
            s0[k].fma(v, a);
            s1[k].fma(v, b);
        }
    }
#pragma unroll
    for (int k = 0; k < PTS; k++) {
        const size_t i = base + (size_t)k * 256;
        if (i < n) {
            out0[i] = s0[k].reduce();
            out1[i] = s1[k].reduce();
        }
    }
}
void launch_linear_combination(const u64 *const *d_col_ptrs, const u64 *d_coefs, unsigned n_cols, size_t n, u64 *d_out0,
                               u64 *d_out1, hipStream_t s) {
    if (n <= ((size_t)1 << 18))
        hipLaunchKernelGGL((linear_combination_kernel<1, 8>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_col_ptrs,
                           d_coefs, n_cols, n, d_out0, d_out1);
    else
        hipLaunchKernelGGL((linear_combination_kernel<4, 1>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s, d_col_ptrs,
                           d_coefs, n_cols, n, d_out0, d_out1);
}

// ---------------------------------------------------------------------------------------------------------
// DEEP:  dst[I] (+)= ( sum_k (coef0_k + coef1_k u) * f_k[I] - (C0 + C1 u) ) / (x_I - at)
// x_I = g * w_N^{bitrev(I)} = 7 * T[I>>1] * (-1)^(I&1).  A thread owns DEEP_PTS points so that one field inversion
// (Montgomery batch trick over the norms of x_I - at) serves all of them.
// ---------------------------------------------------------------------------------------------------------
static constexpr int DEEP_PTS = 4;

__global__ void __launch_bounds__(256)
deep_accumulate_kernel(const u64 *const *cols, const u64 *coefs /*[n][2]*/, unsigned n_cols, size_t N, size_t I0,
                       const u64 *tw, u64 c0, u64 c1, u64 at0, u64 at1, u64 *dst0, u64 *dst1, int accumulate) {
    const size_t base = (size_t)blockIdx.x * (256 * DEEP_PTS) + threadIdx.x;
    Acc160 s0[DEEP_PTS], s1[DEEP_PTS];
#pragma unroll
    for (int k = 0; k < DEEP_PTS; k++) {
        s0[k].clear();
        s1[k].clear();
    }
    for (unsigned c = 0; c < n_cols; c++) {
        const u64 *f = cols[c];
        const u64 a = coefs[2 * c], b = coefs[2 * c + 1];
#pragma unroll
        for (int k = 0; k < DEEP_PTS; k++) {
            size_t I = base + (size_t)k * 256;
            u64 v = I < N ? f[I] : 0;
            s0[k].fma(v, a);
            s1[k].fma(v, b);
        }
    }
    // denominators x_I - at, inverted together
    u64 d0[DEEP_PTS], norm[DEEP_PTS], pref[DEEP_PTS];
    const u64 d1 = gl::neg(at1);
    const u64 seven_d1sq = mul7(gl::sqr(d1));
    u64 run = 1;
#pragma unroll
    for (int k = 0; k < DEEP_PTS; k++) {
        size_t I = base + (size_t)k * 256;
        size_t Ic = I0 + (I < N ? I : 0);   // global LDE index of the point
        u64 wi = tw[Ic >> 1];
        if (Ic & 1) wi = gl::neg(wi);
        u64 x = mul7(wi);
        d0[k] = gl::sub(x, at0);
        norm[k] = gl::sub(gl::sqr(d0[k]), seven_d1sq);
        pref[k] = run;
        run = gl::mul(run, norm[k]);
    }
    u64 inv_run = inv_chain(run);
#pragma unroll
    for (int k = DEEP_PTS - 1; k >= 0; k--) {
        u64 ni = gl::mul(inv_run, pref[k]);
        inv_run = gl::mul(inv_run, norm[k]);
        size_t I = base + (size_t)k * 256;
        if (I < N) {
            gl::e2 den{gl::mul(d0[k], ni), gl::neg(gl::mul(d1, ni))};
            gl::e2 num{gl::sub(s0[k].reduce(), c0), gl::sub(s1[k].reduce(), c1)};
            gl::e2 r = gl::e2_mul(num, den);
            if (accumulate) {
                r.c0 = gl::add(r.c0, gl::canon(dst0[I]));
                r.c1 = gl::add(r.c1, gl::canon(dst1[I]));
            }
            dst0[I] = r.c0;
            dst1[I] = r.c1;
        }
    }
}

// Several opening sets in one pass (the prover's z, z*omega and 0): dst (+)= sum_t (sum_k coef_tk f_tk - C_t) / (x - at_t).
// One inversion serves DEEP_PTS * n_sets denominators instead of DEEP_PTS, and the destination is written once instead of
// being re-read and re-written per set.  Phase 1 walks (set, point) forward building the prefix products of the norms,
// phase 2 walks backward: it unwinds the inverses and computes each set's numerators only then, so that one set's
// accumulators are live at a time.  Exact arithmetic: the same values as deep_accumulate_kernel applied set by set.
struct DeepSetDev {
    const u64 *const *cols;
    const u64 *coefs;       // [n_cols][2]
    unsigned n_cols;
    u64 c0, c1, at0, at1;
};
struct DeepSetsDev {
    DeepSetDev s[DEEP_MAX_SETS];
    int n;
};
__global__ void __launch_bounds__(256)
deep_accumulate_multi_kernel(DeepSetsDev S, size_t N, size_t I0, const u64 *tw, u64 *dst0, u64 *dst1, int accumulate) {
    const size_t base = (size_t)blockIdx.x * (256 * DEEP_PTS) + threadIdx.x;
    u64 x[DEEP_PTS];
#pragma unroll
    for (int k = 0; k < DEEP_PTS; k++) {
        const size_t I = base + (size_t)k * 256;
        const size_t Ic = I0 + (I < N ? I : 0);
        u64 wi = tw[Ic >> 1];
        if (Ic & 1) wi = gl::neg(wi);
        x[k] = mul7(wi);
    }
    u64 norm[DEEP_MAX_SETS][DEEP_PTS], pref[DEEP_MAX_SETS][DEEP_PTS];
    u64 run = 1;
#pragma unroll
    for (int t = 0; t < DEEP_MAX_SETS; t++) {
        if (t >= S.n) break;
        const u64 seven_d1sq = mul7(gl::sqr(S.s[t].at1));   // (-at1)^2 = at1^2
#pragma unroll
        for (int k = 0; k < DEEP_PTS; k++) {
            const u64 d0 = gl::sub(x[k], S.s[t].at0);
            norm[t][k] = gl::sub(gl::sqr(d0), seven_d1sq);
            pref[t][k] = run;
            run = gl::mul(run, norm[t][k]);
        }
    }
    u64 inv_run = inv_chain(run);
    gl::e2 acc[DEEP_PTS];
#pragma unroll
    for (int k = 0; k < DEEP_PTS; k++) {
        const size_t I = base + (size_t)k * 256;
        acc[k] = (accumulate && I < N) ? gl::e2{gl::canon(dst0[I]), gl::canon(dst1[I])} : gl::e2{0, 0};
    }
#pragma unroll
    for (int t = DEEP_MAX_SETS - 1; t >= 0; t--) {
        if (t >= S.n) continue;
        Acc160 s0[DEEP_PTS], s1[DEEP_PTS];
#pragma unroll
        for (int k = 0; k < DEEP_PTS; k++) {
            s0[k].clear();
            s1[k].clear();
        }
        for (unsigned c = 0; c < S.s[t].n_cols; c++) {
            const u64 *f = S.s[t].cols[c];
            const u64 a = S.s[t].coefs[2 * c], b = S.s[t].coefs[2 * c + 1];
#pragma unroll
            for (int k = 0; k < DEEP_PTS; k++) {
                const size_t I = base + (size_t)k * 256;
                const u64 v = f[I < N ? I : 0];
                s0[k].fma(v, a);
                s1[k].fma(v, b);
            }
        }
        const u64 d1 = gl::neg(S.s[t].at1);
#pragma unroll
        for (int k = DEEP_PTS - 1; k >= 0; k--) {
            const u64 ni = gl::mul(inv_run, pref[t][k]);
            inv_run = gl::mul(inv_run, norm[t][k]);
            const u64 d0 = gl::sub(x[k], S.s[t].at0);
            const gl::e2 den{gl::mul(d0, ni), gl::neg(gl::mul(d1, ni))};
            const gl::e2 num{gl::sub(s0[k].reduce(), S.s[t].c0), gl::sub(s1[k].reduce(), S.s[t].c1)};
            acc[k] = gl::e2_add(acc[k], gl::e2_mul(num, den));
        }
    }
#pragma unroll
    for (int k = 0; k < DEEP_PTS; k++) {
        const size_t I = base + (size_t)k * 256;
        if (I < N) {
            dst0[I] = acc[k].c0;
            dst1[I] = acc[k].c1;
        }
    }
}
void launch_deep_accumulate_multi(const DeepSetHostArgs *sets, unsigned n_sets, size_t N, size_t I0, const u64 *d_tw_fwd,
                                  u64 *d_dst0, u64 *d_dst1, int accumulate, hipStream_t s) {
    DeepSetsDev S{};
    S.n = (int)n_sets;
    for (unsigned t = 0; t < n_sets && t < (unsigned)DEEP_MAX_SETS; t++)
        S.s[t] = DeepSetDev{sets[t].d_cols, sets[t].d_coefs, sets[t].n_cols, sets[t].c0, sets[t].c1, sets[t].at0, sets[t].at1};
    const unsigned blocks = (unsigned)((N + 256 * DEEP_PTS - 1) / (256 * DEEP_PTS));
    hipLaunchKernelGGL(deep_accumulate_multi_kernel, dim3(blocks), dim3(256), 0, s, S, N, I0, d_tw_fwd, d_dst0, d_dst1, accumulate);
}

void launch_deep_accumulate(const u64 *const *d_col_ptrs, const u64 *d_coefs, unsigned n_cols, size_t N, size_t I0,
                            const u64 *d_tw_fwd, u64 c0, u64 c1, u64 at0, u64 at1, u64 *d_dst0, u64 *d_dst1,
                            int accumulate, hipStream_t s) {
    unsigned blocks = (unsigned)((N + 256 * DEEP_PTS - 1) / (256 * DEEP_PTS));
    hipLaunchKernelGGL(deep_accumulate_kernel, dim3(blocks), dim3(256), 0, s, d_col_ptrs, d_coefs, n_cols, N, I0,
                       d_tw_fwd, c0, c1, at0, at1, d_dst0, d_dst1, accumulate);
}

}  // namespace bj
