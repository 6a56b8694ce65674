// Prover round 3 on the device: the quotient numerator over the first q cosets of the LDE and its division by the
// vanishing polynomial.
//
// Must equal (as canonical residues), point by point (flat index I = coset*n + i, i bit-reversed):
//   gate evaluation over general purpose columns    src/cs/implementations/prover.rs:1031-1080 with the destination of
//                                                   buffering_source.rs:133-222, 304-362 and the evaluators of
//                                                   cs/gates/{constant_allocator,fma_gate_without_constant,reduction_gate}.rs
//   selectors                                       compute_selector_subpath, prover.rs:2775-2916
//   (z - 1) * L1~                                   prover.rs:1189-1227 with unnormalized_l1_inverse utils.rs:1585-1672
//   copy-permutation chain                          compute_quotient_terms_in_extension copy_permutation.rs:1000-1249
//   lookup terms                                    compute_quotient_terms_for_lookup_specialized lookup_argument_in_ext.rs:949-1319
//   1 / (x^n - 1) per coset                         divide_by_vanishing_for_bitreversed_coset_enumeration utils.rs:770-817
// One lane = one LDE point; consecutive lanes read consecutive addresses of every column (coalesced).  Sums of
// challenge * term products are accumulated unreduced in 160-bit accumulators and reduced once.
#include "gl.h"
#include "kernels.h"

#include <cstdlib>
#include <utility>

using gl::u64;
using gl::u32;

namespace bj {

struct Acc160q {
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

__device__ inline u64 inv_chain3(u64 x) {
    auto sqn = [](u64 v, int n) { for (int i = 0; i < n; i++) v = gl::sqr(v); return v; };
    u64 a1 = x, a2 = gl::mul(sqn(a1, 1), a1), a4 = gl::mul(sqn(a2, 2), a2), a8 = gl::mul(sqn(a4, 4), a4);
    u64 a16 = gl::mul(sqn(a8, 8), a8), a24 = gl::mul(sqn(a16, 8), a8), a28 = gl::mul(sqn(a24, 4), a4);
    u64 a30 = gl::mul(sqn(a28, 2), a2), a31 = gl::mul(sqn(a30, 1), a1);
    u64 b = gl::sqr(a31), a32 = gl::mul(b, x);
    return gl::mul(sqn(b, 32), a32);
}
__device__ __forceinline__ u64 mul7q(u64 a) { return gl::sub(gl::mul_pow2(a, 3), a); }

// x_I = 7 * w_{qn}^{bitrev(I)} = 7 * T[I>>1] * (-1)^(I&1)
__device__ __forceinline__ u64 lde_point(const u64 *tw, size_t I) {
    u64 wi = tw[I >> 1];
    if (I & 1) wi = gl::neg(wi);
    return mul7q(wi);
}

struct GateDev {
    int kind, path_len, reps, var_stride, const_stride, num_terms;
    int path[6];
};
constexpr int BJ_MAX_GATES = 16;   // evaluators over general-purpose columns per circuit (the golden proof's circuit has 11)
struct GateSet {
    GateDev g[BJ_MAX_GATES];
    int n_gates;
};

// T = sum_g selector_g * sum_t alpha_t * term_t            (overwrites out)
__global__ void __launch_bounds__(256)
quotient_gates_kernel(const u64 *vars, size_t var_stride, const u64 *consts, size_t const_stride, GateSet gs,
                      const u64 *alphas /* [n_terms][2] */, size_t Q, u64 *out0, u64 *out1) {
    const size_t I = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= Q) return;
    gl::e2 acc{0, 0};
    int aoff = 0;
    for (int gi = 0; gi < gs.n_gates; gi++) {
        const GateDev &G = gs.g[gi];
        if (G.num_terms == 0) continue;
        if (G.kind >= 5) {   // op-list gate (seam S3) or the Poseidon2 flattened gate: evaluated by their own kernels, only the alpha powers are skipped here
            aoff += G.reps * G.num_terms;
            continue;
        }
        u64 sel = 1;
        for (int b = 0; b < G.path_len; b++) {
            u64 c = gl::canon(consts[(size_t)b * const_stride + I]);
            sel = gl::mul(sel, G.path[b] ? c : gl::sub(1, c));
        }
        Acc160q s0, s1;
        s0.clear();
        s1.clear();
        const size_t cb0 = (size_t)G.path_len;
        u64 k0 = 0, k1 = 0, k2 = 0, k3 = 0;
        if (G.kind == 2) {
            k0 = gl::canon(consts[cb0 * const_stride + I]);
            k1 = gl::canon(consts[(cb0 + 1) * const_stride + I]);
        } else if (G.kind == 3) {
            k0 = gl::canon(consts[cb0 * const_stride + I]);
            k1 = gl::canon(consts[(cb0 + 1) * const_stride + I]);
            k2 = gl::canon(consts[(cb0 + 2) * const_stride + I]);
            k3 = gl::canon(consts[(cb0 + 3) * const_stride + I]);
        }
        for (int r = 0; r < G.reps; r++) {
            const size_t vb = (size_t)r * G.var_stride;
#define VARQ(k) gl::canon(vars[(vb + (k)) * var_stride + I])
            u64 term;
            if (G.kind == 1) {          // ConstantsAllocator: a - c_r
                u64 c = gl::canon(consts[(cb0 + (size_t)r * G.const_stride) * const_stride + I]);
                term = gl::sub(VARQ(0), c);
            } else if (G.kind == 2) {   // FMA: q*a*b + l*c - d
                u64 a = VARQ(0), b = VARQ(1), c = VARQ(2), d = VARQ(3);
                term = gl::sub(gl::add(gl::mul(c, k1), gl::mul(k0, gl::mul(a, b))), d);
            } else {                    // Reduction<4>: sum c_i v_i - r
                Acc160q t;
                t.clear();
                t.fma(VARQ(0), k0);
                t.fma(VARQ(1), k1);
                t.fma(VARQ(2), k2);
                t.fma(VARQ(3), k3);
                term = gl::sub(t.reduce(), VARQ(4));
            }
#undef VARQ
            s0.fma(term, alphas[2 * aoff]);
            s1.fma(term, alphas[2 * aoff + 1]);
            aoff++;
        }
        acc.c0 = gl::add(acc.c0, gl::mul(s0.reduce(), sel));
        acc.c1 = gl::add(acc.c1, gl::mul(s1.reduce(), sel));
    }
    out0[I] = acc.c0;
    out1[I] = acc.c1;
}

// The same sum for the common gate set — at most one gate of each hand-written kind, repetitions 1 / 4 / 5 columns wide —
// with every variable column read ONCE: the columns are taken in windows of 20 (= lcm(4, 5): five FMA repetitions, four
// Reduction<4> repetitions, twenty constant allocations), held in registers, and every gate evaluates its repetitions of the
// window from there.  quotient_gates_kernel above reads the sixty general-purpose columns of the SHA-256 circuit once per
// gate (PMC: 2.05 x the algorithmic bytes); this one reads them once.
constexpr int GW_WINDOW = 20;
struct GateWindows {
    int present[4];         // by kind (1, 2, 3)
    int path_len[4], path[4][6];
    int reps[4], const_stride[4], aoff[4];
    int n_windows, n_vars;
};
__global__ void __launch_bounds__(256)
quotient_gates_windowed_kernel(const u64 *vars, size_t var_stride, const u64 *consts, size_t const_stride, GateWindows gw,
                               const u64 *alphas /* [n_terms][2] */, size_t Q, u64 *out0, u64 *out1) {
    const size_t I = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= Q) return;
    u64 sel[4] = {0, 0, 0, 0};
    u64 pc[6];              // the selector path columns (shared by all gates), read once
#pragma unroll
    for (int b = 0; b < 6; b++) pc[b] = 0;
    {
        int max_path = 0;
#pragma unroll
        for (int k = 1; k <= 3; k++)
            if (gw.present[k] && gw.path_len[k] > max_path) max_path = gw.path_len[k];
#pragma unroll
        for (int b = 0; b < 6; b++)
            if (b < max_path) pc[b] = gl::canon(consts[(size_t)b * const_stride + I]);
    }
#pragma unroll
    for (int k = 1; k <= 3; k++) {
        if (!gw.present[k]) continue;
        u64 sv = 1;
#pragma unroll
        for (int b = 0; b < 6; b++)
            if (b < gw.path_len[k]) sv = gl::mul(sv, gw.path[k][b] ? pc[b] : gl::sub(1, pc[b]));
        sel[k] = sv;
    }
    u64 k2[2] = {0, 0}, k3[4] = {0, 0, 0, 0};
    if (gw.present[2]) {
#pragma unroll
        for (int j = 0; j < 2; j++) k2[j] = gl::canon(consts[(size_t)(gw.path_len[2] + j) * const_stride + I]);
    }
    if (gw.present[3]) {
#pragma unroll
        for (int j = 0; j < 4; j++) k3[j] = gl::canon(consts[(size_t)(gw.path_len[3] + j) * const_stride + I]);
    }
    Acc160q s0[4], s1[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        s0[k].clear();
        s1[k].clear();
    }
#pragma unroll 1
    for (int w = 0; w < gw.n_windows; w++) {
        u64 v[GW_WINDOW];
#pragma unroll
        for (int c = 0; c < GW_WINDOW; c++) {
            const int col = w * GW_WINDOW + c;
            v[c] = gl::canon(vars[(size_t)(col < gw.n_vars ? col : 0) * var_stride + I]);   // past the last column: a valid address, never used
        }
        if (gw.present[1]) {    // ConstantsAllocator: a - c_r
#pragma unroll
            for (int i = 0; i < GW_WINDOW; i++) {
                const int r = w * GW_WINDOW + i;
                if (r < gw.reps[1]) {
                    const u64 c = gl::canon(consts[(size_t)(gw.path_len[1] + r * gw.const_stride[1]) * const_stride + I]);
                    const u64 term = gl::sub(v[i], c);
                    s0[1].fma(term, alphas[2 * (gw.aoff[1] + r)]);
                    s1[1].fma(term, alphas[2 * (gw.aoff[1] + r) + 1]);
                }
            }
        }
        if (gw.present[2]) {    // FMA: q*a*b + l*c - d
#pragma unroll
            for (int i = 0; i < GW_WINDOW / 4; i++) {
                const int r = w * (GW_WINDOW / 4) + i;
                if (r < gw.reps[2]) {
                    const u64 term = gl::sub(gl::add(gl::mul(v[4 * i + 2], k2[1]), gl::mul(k2[0], gl::mul(v[4 * i], v[4 * i + 1]))), v[4 * i + 3]);
                    s0[2].fma(term, alphas[2 * (gw.aoff[2] + r)]);
                    s1[2].fma(term, alphas[2 * (gw.aoff[2] + r) + 1]);
                }
            }
        }
        if (gw.present[3]) {    // Reduction<4>: sum c_i v_i - r
#pragma unroll
            for (int i = 0; i < GW_WINDOW / 5; i++) {
                const int r = w * (GW_WINDOW / 5) + i;
                if (r < gw.reps[3]) {
                    Acc160q t;
                    t.clear();
                    t.fma(v[5 * i], k3[0]);
                    t.fma(v[5 * i + 1], k3[1]);
                    t.fma(v[5 * i + 2], k3[2]);
                    t.fma(v[5 * i + 3], k3[3]);
                    const u64 term = gl::sub(t.reduce(), v[5 * i + 4]);
                    s0[3].fma(term, alphas[2 * (gw.aoff[3] + r)]);
                    s1[3].fma(term, alphas[2 * (gw.aoff[3] + r) + 1]);
                }
            }
        }
    }
    gl::e2 acc{0, 0};
#pragma unroll
    for (int k = 1; k <= 3; k++) {
        if (!gw.present[k]) continue;
        acc.c0 = gl::add(acc.c0, gl::mul(s0[k].reduce(), sel[k]));
        acc.c1 = gl::add(acc.c1, gl::mul(s1[k].reduce(), sel[k]));
    }
    out0[I] = acc.c0;
    out1[I] = acc.c1;
}

// T += sum_i alpha_i * (A_i * (lbeta + sum_j lgamma^j col_ij + lgamma^w tid) - 1) + alpha_B * (B * (lbeta + sum_j lgamma^j tab_j) - mult)
struct LookupQArgs {
    gl::e2 beta;
    gl::e2 gpow[9];
};
__global__ void __launch_bounds__(256)
quotient_lookup_kernel(const u64 *lvars, size_t var_stride, const u64 *table_id, const u64 *tables, size_t tab_stride,
                       const u64 *mult, const u64 *A, const u64 *B, size_t s2_stride, unsigned reps, unsigned w,
                       LookupQArgs la, const u64 *alphas /* [reps+1][2] */, size_t Q, u64 *out0, u64 *out1) {
    const size_t I = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= Q) return;
    Acc160q s0, s1;
    s0.clear();
    s1.clear();
    // table_id == nullptr: the table id is the last of the w + 1 variable columns of every sub-argument
    // (UseSpecializedColumnsWithTableIdAsVariable, lookup_argument_in_ext.rs:949-1000: capacity = w + 1, no constant column)
    const unsigned cps = table_id ? w : w + 1;
    const u64 tid = table_id ? gl::canon(table_id[I]) : 0;
    for (unsigned i = 0; i <= reps; i++) {
        gl::e2 d = la.beta;
        gl::e2 poly;
        u64 minus;
        if (i < reps) {
            for (unsigned j = 0; j < cps; j++) {
                u64 v = gl::canon(lvars[(size_t)(i * cps + j) * var_stride + I]);
                d = gl::e2_add(d, gl::e2_mul_base(la.gpow[j], v));
            }
            if (table_id) d = gl::e2_add(d, gl::e2_mul_base(la.gpow[w], tid));
            poly = {gl::canon(A[((size_t)2 * i) * s2_stride + I]), gl::canon(A[((size_t)2 * i + 1) * s2_stride + I])};
            minus = 1;
        } else {
            for (unsigned j = 0; j <= w; j++) {
                u64 v = gl::canon(tables[(size_t)j * tab_stride + I]);
                d = gl::e2_add(d, gl::e2_mul_base(la.gpow[j], v));
            }
            poly = {gl::canon(B[I]), gl::canon(B[s2_stride + I])};
            minus = gl::canon(mult[I]);
        }
        gl::e2 t = gl::e2_mul(poly, d);
        t.c0 = gl::sub(t.c0, minus);
        // (t0 + t1 u)(a0 + a1 u) = (t0 a0 + 7 t1 a1) + (t0 a1 + t1 a0) u
        const u64 a0 = alphas[2 * i], a1 = alphas[2 * i + 1];
        s0.fma(t.c0, a0);
        s0.fma(t.c1, mul7q(a1));
        s1.fma(t.c0, a1);
        s1.fma(t.c1, a0);
    }
    out0[I] = gl::add(gl::canon(out0[I]), s0.reduce());
    out1[I] = gl::add(gl::canon(out1[I]), s1.reduce());
}

// T = (T + alpha_L1 * (z - 1) * L1~(x) + copy-permutation chain) / (x^n - 1)
struct CopyPermQArgs {
    gl::e2 beta, gamma, alpha_l1;
    u64 xn_minus_one[64];        // per coset: x^n - 1
    u64 vanishing_inv[64];       // per coset: 1 / (x^n - 1)
};
// SMALLK: every non-residue fits 32 bits (always the case for make_non_residues' output; the launcher checks) — the numerator's
// k_c * (x * beta) is then a 32 x 64-bit product per component on top of ONE x * beta per point, instead of two 64 x 64 products
// per column against k_c * beta staged in LDS
template <bool SMALLK>
__global__ void __launch_bounds__(256)
quotient_copy_perm_kernel(const u64 *vars, size_t var_stride, const u64 *sigmas, size_t sig_stride, const u64 *stage2,
                          size_t s2_stride, const u64 *non_res, unsigned V, unsigned chunk, unsigned n_chunks,
                          unsigned log_n, const u64 *tw, CopyPermQArgs ca, const u64 *alphas /* [n_chunks][2] */,
                          size_t Q, size_t I0 /* global index of local point 0 (multi-GPU coset shards) */,
                          const u64 *inv_xm1 /* 1 / (x_I - 1) per local point, or NULL: computed here */, u64 *out0, u64 *out1) {
    // k_c * beta for every column, once per workgroup (a lane would otherwise spend a product per column on k_c * x first)
    extern __shared__ u64 kbeta[];   // [V][2], sized by the launcher (SMALLK: [V] words holding k_c)
    if (SMALLK) {
        for (unsigned t = threadIdx.x; t < V; t += blockDim.x) kbeta[t] = gl::canon(non_res[t]);
    } else {
        for (unsigned t = threadIdx.x; t < 2 * V; t += blockDim.x)
            kbeta[t] = gl::mul(non_res[t >> 1], (t & 1) ? ca.beta.c1 : ca.beta.c0);
    }
    __syncthreads();
    const size_t I = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= Q) return;
    const size_t n = (size_t)1 << log_n;
    const unsigned coset = (unsigned)((I0 + I) >> log_n);   // global coset: selects x^n and the vanishing inverse
    const u32 i_br = (u32)(I & (n - 1));
    const u64 x = lde_point(tw, I0 + I);
    const gl::e2 xb = SMALLK ? gl::e2{gl::mul_weak(x, ca.beta.c0), gl::mul_weak(x, ca.beta.c1)} : gl::e2{0, 0};   // x * beta, once per point
    Acc160q s0, s1;
    s0.clear();
    s1.clear();
    const gl::e2 zv{gl::canon(stage2[I]), gl::canon(stage2[s2_stride + I])};
    {   // (z - 1) * (x^n - 1) / (x - 1) * alpha
        u64 l1 = gl::mul(ca.xn_minus_one[coset], inv_xm1 ? inv_xm1[I] : inv_chain3(gl::sub(x, 1)));
        gl::e2 t{gl::mul(gl::sub(zv.c0, 1), l1), gl::mul(zv.c1, l1)};
        s0.fma(t.c0, ca.alpha_l1.c0);
        s0.fma(t.c1, mul7q(ca.alpha_l1.c1));
        s1.fma(t.c0, ca.alpha_l1.c1);
        s1.fma(t.c1, ca.alpha_l1.c0);
    }
    // z(omega * x): next natural index inside the coset
    const u32 i_next = gl::bitrev32((gl::bitrev32(i_br, log_n) + 1) & (u32)(n - 1), log_n);
    const size_t In = (I - i_br) + i_next;
    const gl::e2 z_shift{stage2[In], stage2[s2_stride + In]};
    // The two product chains of a chunk run on WEAK residues (gl::mul_weak / add_weak: any u64 congruent to the value): raw
    // words from memory in, nothing canonicalised until the lazy accumulators take lhs - rhs — a chain link is four weak
    // products and two weak sums per F_p^2 factor instead of Karatsuba's three canonical products and five canonical sums.
    for (unsigned j = 0; j < n_chunks; j++) {
        gl::e2 lhs = (j + 1 < n_chunks) ? gl::e2{stage2[((size_t)2 + 2 * j) * s2_stride + I], stage2[((size_t)3 + 2 * j) * s2_stride + I]} : z_shift;
        gl::e2 rhs = (j == 0) ? zv : gl::e2{stage2[((size_t)2 * j) * s2_stride + I], stage2[((size_t)2 * j + 1) * s2_stride + I]};
        for (unsigned c = j * chunk; c < (j + 1) * chunk && c < V; c++) {
            const u64 wg = gl::add_weak(vars[(size_t)c * var_stride + I], ca.gamma.c0);   // w + gamma_0, shared by both factors
            const u64 sg = sigmas[(size_t)c * sig_stride + I];
            const gl::e2 d{gl::add_weak(gl::mul_weak(sg, ca.beta.c0), wg), gl::add_weak(gl::mul_weak(sg, ca.beta.c1), ca.gamma.c1)};
            lhs = gl::e2_mul_weak(lhs, d);
            gl::e2 nm;
            if (SMALLK) {
                const u32 k = (u32)kbeta[c];
                nm = {gl::add_weak(gl::mul_u32_weak(xb.c0, k), wg), gl::add_weak(gl::mul_u32_weak(xb.c1, k), ca.gamma.c1)};
            } else {
                nm = {gl::add_weak(gl::mul_weak(x, kbeta[2 * c]), wg), gl::add_weak(gl::mul_weak(x, kbeta[2 * c + 1]), ca.gamma.c1)};
            }
            rhs = gl::e2_mul_weak(rhs, nm);
        }
        const gl::e2 t{gl::sub_weak(lhs.c0, rhs.c0), gl::sub_weak(lhs.c1, rhs.c1)};
        const u64 a0 = alphas[2 * j], a1 = alphas[2 * j + 1];
        s0.fma(t.c0, a0);
        s0.fma(t.c1, mul7q(a1));
        s1.fma(t.c0, a1);
        s1.fma(t.c1, a0);
    }
    u64 r0 = gl::add(gl::canon(out0[I]), s0.reduce());
    u64 r1 = gl::add(gl::canon(out1[I]), s1.reduce());
    const u64 vi = ca.vanishing_inv[coset];
    out0[I] = gl::mul(r0, vi);
    out1[I] = gl::mul(r1, vi);
}

// 1 / (x_I - 1) for the points I0 .. I0 + Q of the LDE domain: depends on the domain only, so a setup computes it once (the
// fixed exponentiation is ~75 products per point, 3 % of quotient_copy_perm's arithmetic at 92 columns)
__global__ void __launch_bounds__(256) inv_x_minus_one_kernel(const u64 *tw, size_t Q, size_t I0, u64 *out) {
    const size_t I = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I < Q) out[I] = inv_chain3(gl::sub(lde_point(tw, I0 + I), 1));
}

// Sharded quotient: rank i has evaluated the quotient terms on the first E = q n / W points of its own LDE range
// — the coset s_i * H_E — and inverse-transformed them: R_i = T mod (x^E - a_i), a_i = s_i^E.  With T = sum_j x^(jE) T_j
// (deg T_j < E), R_i = sum_j a_i^j T_j: the W x W Vandermonde system is solved coefficient by coefficient with the host's
// inverse matrix.  residues: [W][n_cols][E] (the all-gathered blocks), out: [n_cols][W * E], canonical.
struct CombineArgs {
    u64 vinv[64];    // [j][i], row-major, W <= 8
};
__global__ void __launch_bounds__(256)
combine_residues_kernel(const u64 *__restrict__ residues, unsigned W, size_t E, unsigned n_cols, CombineArgs ca, u64 *__restrict__ out) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned col = blockIdx.y;
    if (k >= E) return;
    u64 r[8];
    for (unsigned i = 0; i < W; i++) r[i] = residues[((size_t)i * n_cols + col) * E + k];
    for (unsigned j = 0; j < W; j++) {
        Acc160q acc;
        acc.clear();
        for (unsigned i = 0; i < W; i++) acc.fma(r[i], ca.vinv[j * W + i]);
        out[(size_t)col * W * E + (size_t)j * E + k] = acc.reduce();
    }
}

// ----------------------------------------------------------------------------------------------- launchers
// h_a[i] = a_i (canonical, pairwise distinct); returns false if the Vandermonde matrix is singular or W > 8
bool launch_combine_residues(const u64 *d_residues, unsigned W, size_t E, unsigned n_cols, const u64 *h_a, u64 *d_out, hipStream_t s) {
    if (W == 0 || W > 8 || !E || !n_cols) return false;
    // inverse of V[i][j] = a_i^j by Gauss-Jordan over F_p
    u64 A[8][16];
    for (unsigned i = 0; i < W; i++) {
        u64 p = 1;
        for (unsigned j = 0; j < W; j++) {
            A[i][j] = p;
            p = gl::mul(p, gl::canon(h_a[i]));
            A[i][W + j] = i == j ? 1 : 0;
        }
    }
    for (unsigned c = 0; c < W; c++) {
        unsigned piv = c;
        while (piv < W && A[piv][c] == 0) piv++;
        if (piv == W) return false;
        if (piv != c)
            for (unsigned k = 0; k < 2 * W; k++) std::swap(A[c][k], A[piv][k]);
        const u64 inv = gl::inv(A[c][c]);
        for (unsigned k = 0; k < 2 * W; k++) A[c][k] = gl::mul(A[c][k], inv);
        for (unsigned r = 0; r < W; r++) {
            if (r == c || A[r][c] == 0) continue;
            const u64 f = A[r][c];
            for (unsigned k = 0; k < 2 * W; k++) A[r][k] = gl::sub(A[r][k], gl::mul(f, A[c][k]));
        }
    }
    CombineArgs ca{};
    for (unsigned j = 0; j < W; j++)
        for (unsigned i = 0; i < W; i++) ca.vinv[j * W + i] = A[j][W + i];
    hipLaunchKernelGGL(combine_residues_kernel, dim3((unsigned)((E + 255) / 256), n_cols), dim3(256), 0, s, d_residues, W, E, n_cols, ca, d_out);
    return true;
}

void launch_inv_x_minus_one(const u64 *d_tw_fwd, size_t Q, size_t I0, u64 *d_out, hipStream_t s) {
    if (Q) hipLaunchKernelGGL(inv_x_minus_one_kernel, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, s, d_tw_fwd, Q, I0, d_out);
}

void launch_quotient_gates(const u64 *d_vars, size_t var_stride, const u64 *d_consts, size_t const_stride,
                           const int *h_gates_flat /* 12 ints per gate */, unsigned n_gates, const u64 *d_alphas,
                           size_t Q, u64 *d_out0, u64 *d_out1, hipStream_t s) {
    GateSet gs;
    gs.n_gates = (int)n_gates;
    for (unsigned g = 0; g < n_gates && g < (unsigned)BJ_MAX_GATES; g++) {
        const int *f = h_gates_flat + 12 * g;
        gs.g[g] = GateDev{f[0], f[1], f[2], f[3], f[4], f[5], {f[6], f[7], f[8], f[9], f[10], f[11]}};
    }
    {   // the windowed kernel when the hand-written kinds appear at most once each with their principal widths
        const bool windowed_on = bj::env().gates_windowed;   // BJ_GATES_WINDOWED=0 selects the per-gate kernel (the tests run both)
        GateWindows gw{};
        const int width[4] = {0, 1, 4, 5};
        bool ok = windowed_on && n_gates <= (unsigned)BJ_MAX_GATES;
        int aoff = 0, span = 0, n_hand = 0;
        for (unsigned g = 0; ok && g < n_gates; g++) {
            const GateDev &G = gs.g[g];
            if (G.num_terms == 0) continue;
            if (G.kind >= 5) {   // evaluated by its own kernel: only the alpha powers are skipped (as in quotient_gates_kernel)
                aoff += G.reps * G.num_terms;
                continue;
            }
            if (G.kind < 1 || G.kind > 3 || gw.present[G.kind] || G.var_stride != width[G.kind] || G.num_terms != 1 || G.path_len > 6 ||
                (G.kind != 1 && G.const_stride != 0)) {
                ok = false;
                break;
            }
            gw.present[G.kind] = 1;
            gw.path_len[G.kind] = G.path_len;
            for (int b = 0; b < 6; b++) gw.path[G.kind][b] = G.path[b];
            gw.reps[G.kind] = G.reps;
            gw.const_stride[G.kind] = G.const_stride;
            gw.aoff[G.kind] = aoff;
            aoff += G.reps;
            if (G.reps * width[G.kind] > span) span = G.reps * width[G.kind];
            n_hand++;
        }
        if (ok && n_hand > 0) {
            gw.n_windows = (span + GW_WINDOW - 1) / GW_WINDOW;
            gw.n_vars = span;
            hipLaunchKernelGGL(quotient_gates_windowed_kernel, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, s, d_vars, var_stride,
                               d_consts, const_stride, gw, d_alphas, Q, d_out0, d_out1);
            return;
        }
    }
    hipLaunchKernelGGL(quotient_gates_kernel, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, s, d_vars, var_stride,
                       d_consts, const_stride, gs, d_alphas, Q, d_out0, d_out1);
}

void launch_quotient_lookup(const u64 *d_lvars, size_t var_stride, const u64 *d_table_id, const u64 *d_tables,
                            size_t tab_stride, const u64 *d_mult, const u64 *d_A, const u64 *d_B, size_t s2_stride,
                            unsigned reps, unsigned w, const u64 *lbeta, const u64 *lgamma, const u64 *d_alphas,
                            size_t Q, u64 *d_out0, u64 *d_out1, hipStream_t s) {
    LookupQArgs la;
    la.beta = {gl::canon(lbeta[0]), gl::canon(lbeta[1])};
    gl::e2 g{gl::canon(lgamma[0]), gl::canon(lgamma[1])};
    la.gpow[0] = {1, 0};
    for (unsigned j = 1; j <= w && j < 9; j++) la.gpow[j] = gl::e2_mul(la.gpow[j - 1], g);
    hipLaunchKernelGGL(quotient_lookup_kernel, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, s, d_lvars, var_stride,
                       d_table_id, d_tables, tab_stride, d_mult, d_A, d_B, s2_stride, reps, w, la, d_alphas, Q, d_out0,
                       d_out1);
}

void launch_quotient_copy_perm(const u64 *d_vars, size_t var_stride, const u64 *d_sigmas, size_t sig_stride,
                               const u64 *d_stage2, size_t s2_stride, const u64 *d_non_res, unsigned V, unsigned chunk,
                               unsigned log_n, unsigned log_L, const u64 *d_tw_fwd, const u64 *beta, const u64 *gamma,
                               const u64 *alpha_l1, const u64 *d_alphas_cp, size_t Q_local, size_t I0, const u64 *d_inv_xm1, u64 *d_out0,
                               u64 *d_out1, hipStream_t s, bool small_non_residues) {
    const size_t n = (size_t)1 << log_n, Q = Q_local;
    const unsigned L = 1u << log_L;   // cosets of the whole LDE domain; a GPU may hold any contiguous range of them
    CopyPermQArgs ca;
    ca.beta = {gl::canon(beta[0]), gl::canon(beta[1])};
    ca.gamma = {gl::canon(gamma[0]), gl::canon(gamma[1])};
    ca.alpha_l1 = {gl::canon(alpha_l1[0]), gl::canon(alpha_l1[1])};
    // x^n on coset c: (7 * w_{Ln}^{bitrev_L(c)})^n = 7^n * w_L^{bitrev_L(c)}   (for c < q this is w_q^{bitrev_q(c)})
    const u64 g_n = gl::pow(gl::GEN, n);
    const u64 wq = gl::omega(log_L);
    for (unsigned c = 0; c < 64; c++) {
        if (c < L) {
            u64 xn = gl::mul(g_n, gl::pow(wq, gl::bitrev32(c, log_L)));
            ca.xn_minus_one[c] = gl::sub(xn, 1);
            ca.vanishing_inv[c] = gl::inv(ca.xn_minus_one[c]);
        } else {
            ca.xn_minus_one[c] = 0;
            ca.vanishing_inv[c] = 0;
        }
    }
    const unsigned n_chunks = (V + chunk - 1) / chunk;
    if (small_non_residues && !env().copy_perm_wide_k)
        hipLaunchKernelGGL(quotient_copy_perm_kernel<true>, dim3((unsigned)((Q + 255) / 256)), dim3(256), (size_t)V * 8, s, d_vars, var_stride,
                           d_sigmas, sig_stride, d_stage2, s2_stride, d_non_res, V, chunk, n_chunks, log_n, d_tw_fwd, ca,
                           d_alphas_cp, Q, I0, d_inv_xm1, d_out0, d_out1);
    else
        hipLaunchKernelGGL(quotient_copy_perm_kernel<false>, dim3((unsigned)((Q + 255) / 256)), dim3(256), (size_t)V * 16, s, d_vars, var_stride,
                           d_sigmas, sig_stride, d_stage2, s2_stride, d_non_res, V, chunk, n_chunks, log_n, d_tw_fwd, ca,
                           d_alphas_cp, Q, I0, d_inv_xm1, d_out0, d_out1);
}

// ----------------------------------------------------------------------------------------------- query gathers
// out[q][c] = base[c * col_stride + idx[q]]   (leaf elements of a base oracle, proof.rs:65-100)
__global__ void gather_rows_kernel(const u64 *base, size_t col_stride, unsigned n_cols, const u64 *idx, unsigned n_idx,
                                   u64 *out) {
    unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned qi = blockIdx.y;
    if (c >= n_cols || qi >= n_idx) return;
    out[(size_t)qi * n_cols + c] = gl::canon(base[(size_t)c * col_stride + idx[qi]]);
}
void launch_gather_rows(const u64 *d_base, size_t col_stride, unsigned n_cols, const u64 *d_idx, unsigned n_idx,
                        u64 *d_out, hipStream_t s) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3((n_cols + 63) / 64, n_idx), dim3(64), 0, s, d_base, col_stride, n_cols,
                       d_idx, n_idx, d_out);
}
// Merkle paths for many leaves at once: out[q][d][4] = sibling at depth d  (merkle_tree.rs:462-480)
__global__ void merkle_paths_kernel(const u64 *tree, size_t num_leaves, unsigned depth, const u64 *idx, unsigned n_idx,
                                    u64 *out) {
    unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_idx * depth) return;
    unsigned qi = t / depth, d = t % depth;
    size_t off = 0, len = num_leaves;
    for (unsigned k = 0; k < d; k++) {
        off += len;
        len >>= 1;
    }
    size_t pos = (idx[qi] >> d) ^ 1;
    const u64 *src = tree + 4 * (off + pos);
    u64 *dst = out + ((size_t)qi * depth + d) * 4;
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
}
void launch_merkle_paths(const u64 *d_tree, size_t num_leaves, unsigned depth, const u64 *d_idx, unsigned n_idx,
                         u64 *d_out, hipStream_t s) {
    unsigned total = n_idx * depth;
    if (!total) return;
    hipLaunchKernelGGL(merkle_paths_kernel, dim3((total + 63) / 64), dim3(64), 0, s, d_tree, num_leaves, depth, d_idx,
                       n_idx, d_out);
}

// FRI leaf = 2^k values of c0 then 2^k values of c1 at leaf index j (merkle_tree.rs:285-292)
__global__ void gather_fri_leaves_kernel(const u64 *c0, const u64 *c1, unsigned log_e, const u64 *leaf_idx, unsigned n_idx,
                                         u64 *out) {
    const unsigned E = 1u << log_e;
    unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_idx * 2 * E) return;
    unsigned qi = t / (2 * E), e = t % (2 * E);
    const u64 *src = e < E ? c0 : c1;
    out[t] = gl::canon(src[leaf_idx[qi] * E + (e & (E - 1))]);
}
void launch_gather_fri_leaves(const u64 *d_c0, const u64 *d_c1, unsigned log_e, const u64 *d_leaf_idx, unsigned n_idx,
                              u64 *d_out, hipStream_t s) {
    unsigned total = n_idx * (2u << log_e);
    if (!total) return;
    hipLaunchKernelGGL(gather_fri_leaves_kernel, dim3((total + 63) / 64), dim3(64), 0, s, d_c0, d_c1, log_e, d_leaf_idx,
                       n_idx, d_out);
}

}  // namespace bj
