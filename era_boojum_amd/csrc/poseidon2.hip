// Poseidon2 (Goldilocks, t = 12, rate 8 / capacity 4, x^7, 4 + 22 + 4 rounds) Merkle-tree hashing for gfx950.
//
// Must equal the reference's CPU tree hasher bit for bit (as canonical residues):
//   permutation        src/implementations/poseidon2/state_generic_impl.rs:128-233
//   external matrix    src/implementations/suggested_mds.rs:21-103      circ(2*M4, M4, M4)
//   internal matrix    src/implementations/poseidon2/params.rs:38-39    1 + diag(2^{4,14,11,8,0,5,2,9,13,6,3,12})
//   sponge             src/algebraic_props/sponge.rs:224-346            overwrite absorption, zero-padded tail, no length tag
//   leaf / node hash   src/cs/oracle/mod.rs:114-176
//   tree               src/cs/oracle/merkle_tree.rs:78-174 (construct), 176-386 (chunked), 388-449 (node layers)
//
// Mapping: one lane = one leaf (or one parent node).  The 12-word sponge state lives in VGPRs for the whole leaf;
// for leaf I the lane reads element I of every column, so a wavefront reads 64 consecutive u64 of one column per
// load (512 B, coalesced).  Round constants are wave-uniform and come through the scalar cache.
// The work is integer-ALU bound (~472 field multiplications per 64 absorbed bytes), not HBM bound.
#include "gl.h"
#include "kernels.h"
#include "poseidon_rc.inc"
#include <cstdlib>

using gl::u64;
using gl::u32;

namespace bj {

__constant__ u64 POSEIDON_RC[BJ_POSEIDON_NUM_RC] = BJ_POSEIDON_RC_TABLE;

// ---------------------------------------------------------------------------------------------------------
// Arithmetic inside the permutation is LAZY: state words are "weak" residues (any u64 congruent to the value, not
// necessarily < p) and the linear layers accumulate in 96-bit integers (three 32-bit words) that are folded back with
// 2^64 = 2^32 - 1 only once per output.  Every step is exact mod p, so the canonicalised digest equals the reference's;
// it removes ~18 % of the VALU instructions of the canonical formulation (the kernel is VALU-bound).
// Invariants (proved in the comments below): a weak value is < 2^64; a correction "+EPS on carry" never carries twice.
// ---------------------------------------------------------------------------------------------------------
struct W3 {
    u32 w0, w1, w2;
};
__device__ __forceinline__ W3 w3_from(u64 a) { return {gl::lo32(a), gl::hi32(a), 0u}; }
__device__ __forceinline__ W3 w3_add(W3 a, W3 b) {
    u32 c, d;
    W3 r;
    r.w0 = __builtin_addc(a.w0, b.w0, 0u, &c);
    r.w1 = __builtin_addc(a.w1, b.w1, c, &c);
    r.w2 = __builtin_addc(a.w2, b.w2, c, &d);
    return r;
}
__device__ __forceinline__ W3 w3_add64(W3 a, u64 b) {
    u32 c, d;
    W3 r;
    r.w0 = __builtin_addc(a.w0, gl::lo32(b), 0u, &c);
    r.w1 = __builtin_addc(a.w1, gl::hi32(b), c, &c);
    r.w2 = __builtin_addc(a.w2, 0u, c, &d);
    return r;
}
__device__ __forceinline__ W3 w3_sum64(u64 a, u64 b) {   // a + b as a 65-bit integer
    u32 c, d;
    W3 r;
    r.w0 = __builtin_addc(gl::lo32(a), gl::lo32(b), 0u, &c);
    r.w1 = __builtin_addc(gl::hi32(a), gl::hi32(b), c, &c);
    r.w2 = __builtin_addc(0u, 0u, c, &d);
    return r;
}
template <unsigned K>
__device__ __forceinline__ W3 w3_shl(W3 a) {   // a * 2^K, 0 < K < 32; the caller guarantees no overflow of 96 bits
    return {a.w0 << K, __builtin_amdgcn_alignbit(a.w1, a.w0, 32 - K), __builtin_amdgcn_alignbit(a.w2, a.w1, 32 - K)};
}
template <unsigned K>
__device__ __forceinline__ W3 w3_shl64(u64 a) {   // a * 2^K as a 96-bit integer, 0 < K < 32
    const u32 lo = gl::lo32(a), hi = gl::hi32(a);
    return {lo << K, __builtin_amdgcn_alignbit(hi, lo, 32 - K), hi >> (32 - K)};
}
// lo + m * (2^32 - 1) as ONE v_mad_u64_u32; e = EPS where the 64-bit sum wrapped, else 0.  Carry-writing VALU ops
// (v_add_co/v_addc_co/v_sub_co/...) cost ~2x a plain op on gfx950 and each carry link needs 2 wait states, so one
// multiply-add replaces the (m << 32) - m construction and its 64-bit add (4 carry ops).  The s_nop covers the
// VALU-SGPR-write -> VALU-read wait states for the mask (the compiler cannot see inside the string).
__device__ __forceinline__ u64 mad_eps(u32 m, u64 lo, u32 &e) {
    u64 r, cm;
    asm("v_mad_u64_u32 %[r], %[cm], %[m], -1, %[lo]\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32 %[e], 0, -1, %[cm]"
        : [r] "=&v"(r), [cm] "=&s"(cm), [e] "=v"(e)
        : [m] "v"(m), [lo] "v"(lo));
    return r;
}
// 96-bit integer with w2 < 2^31 -> weak residue:  (w1:w0) + w2 * (2^32 - 1), "+EPS" once on carry.
// No second carry: after a wrap the sum is < w2 * 2^32 <= 2^63, far below 2^64 - EPS.
__device__ __forceinline__ u64 w3_reduce(W3 a) {
    u32 e;
    u64 r = mad_eps(a.w2, gl::pack(a.w0, a.w1), e);
    return r + (u64)e;
}
// weak x weak -> weak (same limbs as gl::mul, reduction without the final canonicalisation):
//   V = lo + hl*EPS - hh  in (-2^32, 2^65 - 2^33);   R = lo + hl*EPS mod 2^64 (carry c), then R - hh mod 2^64 (borrow b)
//   c only: true value = R + 2^64 = R + EPS (R < 2^64 - 2^33, no second carry);  b only: R - EPS (R > 2^64 - 2^32);
//   both: the wrap and the borrow cancel.  So W = R + (c ? EPS : 0) - (b ? EPS : 0) mod 2^64 in every case.
__device__ __forceinline__ u64 mulw(u64 a, u64 b) {
#if !defined(BJ_P2_MULW_LIMBS)
    return gl::mul_weak(a, b);   // chained multiply-adds, 12 instructions (gl.h)
#else
    u32 hh, hl, e;
    u64 lo;
    gl::mul_limbs(a, b, hh, hl, lo);
    u64 r = mad_eps(hl, lo, e);
    // r - hh with the borrow chain written out: the compiler lowers the C borrow idiom to cndmask + sub pairs (8
    // instructions for what is sub, subb, cndmask); the s_nops are the wait states between carry producer and consumer
    u32 d0, d1, f;
    asm("v_sub_co_u32 %[d0], vcc, %[r0], %[hh]\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %[d1], vcc, 0, %[r1], vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32 %[f], 0, -1, vcc"
        : [d0] "=&v"(d0), [d1] "=&v"(d1), [f] "=v"(f)
        : [r0] "v"(gl::lo32(r)), [r1] "v"(gl::hi32(r)), [hh] "v"(hh)
        : "vcc");
    return gl::pack(d0, d1) + (u64)e - (u64)f;
#endif
}
// weak + canonical constant -> weak: "+EPS" on carry; the wrapped sum is < rc < p, so adding EPS cannot carry again
__device__ __forceinline__ u64 addw_rc(u64 x, u64 rc) {
    u32 c1, c2, c3, c4;
    u32 s0 = __builtin_addc(gl::lo32(x), gl::lo32(rc), 0u, &c1);
    u32 s1 = __builtin_addc(gl::hi32(x), gl::hi32(rc), c1, &c2);
    u32 e = c2 ? 0xFFFFFFFFu : 0u;
    s0 = __builtin_addc(s0, e, 0u, &c3);
    s1 = __builtin_addc(s1, 0u, c3, &c4);
    return gl::pack(s0, s1);
}
__device__ __forceinline__ u64 pow7w(u64 x) {
    u64 x2 = mulw(x, x), x3 = mulw(x2, x), x4 = mulw(x2, x2);
    return mulw(x4, x3);
}

// M4 block [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]] on 64-bit inputs, 96-bit outputs (row sums <= 16 -> < 2^68)
__device__ __forceinline__ void m4w(u64 x0, u64 x1, u64 x2, u64 x3, W3 &y0, W3 &y1, W3 &y2, W3 &y3) {
    W3 t0 = w3_sum64(x0, x1), t1 = w3_sum64(x2, x3);
    W3 t2 = w3_add(w3_shl64<1>(x1), t1), t3 = w3_add(w3_shl64<1>(x3), t0);
    W3 t4 = w3_add(w3_shl<2>(t1), t3), t5 = w3_add(w3_shl<2>(t0), t2);
    y0 = w3_add(t3, t5);
    y1 = t5;
    y2 = w3_add(t2, t4);
    y3 = t4;
}
// external matrix circ(2*M4, M4, M4): out[4b+j] = y_b[j] + sum_b' y_b'[j]; coefficients sum to <= 64 -> < 2^70
__device__ __forceinline__ void ext_mds(u64 (&s)[12]) {
    W3 y[12];
    m4w(s[0], s[1], s[2], s[3], y[0], y[1], y[2], y[3]);
    m4w(s[4], s[5], s[6], s[7], y[4], y[5], y[6], y[7]);
    m4w(s[8], s[9], s[10], s[11], y[8], y[9], y[10], y[11]);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        W3 sum = w3_add(w3_add(y[j], y[4 + j]), y[8 + j]);
        s[j] = w3_reduce(w3_add(y[j], sum));
        s[4 + j] = w3_reduce(w3_add(y[4 + j], sum));
        s[8 + j] = w3_reduce(w3_add(y[8 + j], sum));
    }
}

// a * 2^K + sum with multiply-adds instead of shift / 96-bit add / fold (9 -> 6 instructions, 4 -> 1 carry op):
//   a = a0 + a1*2^32,  sum = sl + sh*2^32 with sl, sh < 2^36 (sums of the low / high words of the state)
//   A = a0*2^K + sl                < 2^47          (no carry)
//   B = a1*2^K + sh                < 2^47
//   a*2^K + sum = A + B*2^32 = A + B_hi*2^64 + B_lo*2^32  ==  A + B_hi*EPS + B_lo*2^32      (B_hi < 2^15)
//   T = A + B_hi*EPS < 2^48 (no carry);  result = T + (B_lo << 32): one add on the high word, "+EPS" on its carry
//   (after a wrap the value is < 2^48, so the correction cannot carry again).
// The power of two travels in an SGPR (VOP3 has no literal operand on gfx950).
__device__ __forceinline__ u64 mad64(u32 x, u32 y, u64 c) {   // x*y + c, carry-out discarded (callers prove there is none)
    u64 r;
    asm("v_mad_u64_u32 %[r], vcc, %[x], %[y], %[c]" : [r] "=v"(r) : [x] "v"(x), [y] "s"(y), [c] "v"(c) : "vcc");
    return r;
}
__device__ __forceinline__ u64 acc32(u32 x, u64 c) {   // c + x (zero-extended), c < 2^63
    u64 r;
    asm("v_mad_u64_u32 %[r], vcc, %[x], 1, %[c]" : [r] "=v"(r) : [x] "v"(x), [c] "v"(c) : "vcc");
    return r;
}
// A + B * 2^32 -> weak residue, for A < 2^62 and B < 2^62 with A + (B >> 32) * EPS < 2^63 (no carry):
//   B * 2^32 = B_hi * 2^64 + B_lo * 2^32 == B_hi * EPS + B_lo * 2^32;  T = A + B_hi * EPS;  result = T + (B_lo << 32): one add on
//   the high word, "+EPS" on its carry (after a wrap the value is < 2^63, so the correction cannot carry again).
__device__ __forceinline__ u64 combine_split(u64 A, u64 B) {
    u64 T;
    asm("v_mad_u64_u32 %[t], vcc, %[b], -1, %[a]" : [t] "=v"(T) : [b] "v"(gl::hi32(B)), [a] "v"(A) : "vcc");
    u32 c, hi, e;
    hi = __builtin_addc(gl::hi32(T), gl::lo32(B), 0u, &c);
    e = c ? 0xFFFFFFFFu : 0u;
    return gl::pack(gl::lo32(T), hi) + (u64)e;
}
template <unsigned K>
__device__ __forceinline__ u64 shl_plus(u64 a, u64 sum_lo, u64 sum_hi) {
    const u64 A = K ? mad64(gl::lo32(a), 1u << K, sum_lo) : acc32(gl::lo32(a), sum_lo);
    const u64 B = K ? mad64(gl::hi32(a), 1u << K, sum_hi) : acc32(gl::hi32(a), sum_hi);
    return combine_split(A, B);
}

}  // namespace bj
#ifndef BJ_P2_ASM_INC     // tools/p2_variants.py builds the same library around other schedules of the stream
#define BJ_P2_ASM_INC "p2_asm.inc"
#endif
#include BJ_P2_ASM_INC    // generated (tools/gen_p2_asm.py): poseidon2_permutation_asm, the scheduled instruction stream
namespace bj {

// state in: any u64 words; state out: weak words (canonicalise what leaves the sponge with gl::canon)
__device__ __forceinline__ void poseidon2_permutation(u64 (&s)[12]) {
#if !defined(BJ_P2_CPP)
    poseidon2_permutation_asm(s);
    return;
#endif
    ext_mds(s);
    int r = 0;
#pragma unroll 1
    for (int i = 0; i < 4; i++, r++) {
#pragma unroll
        for (int k = 0; k < 12; k++) s[k] = pow7w(addw_rc(s[k], POSEIDON_RC[12 * r + k]));
        ext_mds(s);
    }
#pragma unroll 1
    for (int i = 0; i < 22; i++, r++) {
        s[0] = pow7w(addw_rc(s[0], POSEIDON_RC[12 * r]));
        // sum of the state as two carry-free accumulators: sl = sum of the low words, sh = sum of the high words (< 2^36
        // each), one multiply-add per word instead of a 96-bit carry chain;  sum = sl + sh * 2^32
        u64 sl = (u64)gl::lo32(s[0]), sh = (u64)gl::hi32(s[0]);
#pragma unroll
        for (int k = 1; k < 12; k++) {
            sl = acc32(gl::lo32(s[k]), sl);
            sh = acc32(gl::hi32(s[k]), sh);
        }
        // internal matrix 1 + diag(2^{4,14,11,8,0,5,2,9,13,6,3,12})  (poseidon2/params.rs:38-39)
        s[0] = shl_plus<4>(s[0], sl, sh);
        s[1] = shl_plus<14>(s[1], sl, sh);
        s[2] = shl_plus<11>(s[2], sl, sh);
        s[3] = shl_plus<8>(s[3], sl, sh);
        s[4] = shl_plus<0>(s[4], sl, sh);
        s[5] = shl_plus<5>(s[5], sl, sh);
        s[6] = shl_plus<2>(s[6], sl, sh);
        s[7] = shl_plus<9>(s[7], sl, sh);
        s[8] = shl_plus<13>(s[8], sl, sh);
        s[9] = shl_plus<6>(s[9], sl, sh);
        s[10] = shl_plus<3>(s[10], sl, sh);
        s[11] = shl_plus<12>(s[11], sl, sh);
    }
#pragma unroll 1
    for (int i = 0; i < 4; i++, r++) {
#pragma unroll
        for (int k = 0; k < 12; k++) s[k] = pow7w(addw_rc(s[k], POSEIDON_RC[12 * r + k]));
        ext_mds(s);
    }
}

// ---------------------------------------------------------------------------------------------------------
// leaf hashing: leaf I = sponge(cols[0][I], cols[1][I], ...)             (merkle_tree.rs:78-174)
// cols given either as base + c*stride or through a device array of column pointers.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
poseidon2_leaves_kernel(const u64 *base, size_t col_stride, const u64 *const *col_ptrs, unsigned n_cols,
                        size_t num_leaves, u64 *digests) {
    size_t I = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= num_leaves) return;
    u64 s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = 0;
    // ONE call site of the permutation: with a second copy for the zero-padded tail block the kernel is 72 KB of code,
    // more than the 64 KB instruction cache two CUs share; the tail's zeros are selected by wave-uniform conditions
    for (unsigned c = 0; c < n_cols; c += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (c + k < n_cols) {
                const u64 *p = col_ptrs ? col_ptrs[c + k] : base + (size_t)(c + k) * col_stride;
                s[k] = p[I];
            } else {
                s[k] = 0;
            }
        }
        poseidon2_permutation(s);
    }
    // digest = state[0..4]; 32 B per lane
    ulonglong2 *d = reinterpret_cast<ulonglong2 *>(digests + 4 * I);
    d[0] = make_ulonglong2(gl::canon(s[0]), gl::canon(s[1]));
    d[1] = make_ulonglong2(gl::canon(s[2]), gl::canon(s[3]));
}

// A RUN of absorptions of the same sponge, for leaves whose columns arrive in groups (bj_prove: the witness comes over PCIe
// while the first groups are already being extended and hashed): state[8..12] <- what the previous group left in `capacity`
// ([4][num_leaves], zeros before the first group); then, eight columns at a time, state[0..8] <- the group's next elements
// (zero-padded in the last block of the last group), permute — every group but the last holds a multiple of eight columns;
// the last group writes the digest, the others their capacity words.  Group by group this is exactly poseidon2_leaves_kernel's
// loop (sponge.rs:224-346: overwrite mode, no length tag), with ONE call site of the permutation as there.
__global__ void __launch_bounds__(256)
poseidon2_leaves_absorb_kernel(const u64 *base, size_t col_stride, unsigned n_cols, size_t num_leaves, u64 *capacity,
                               u64 *digests, int first, int last) {
    size_t I = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= num_leaves) return;
    u64 s[12];
#pragma unroll
    for (int k = 0; k < 4; k++) s[8 + k] = first ? 0 : capacity[(size_t)k * num_leaves + I];
    for (unsigned c = 0; c < n_cols; c += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) s[k] = c + k < n_cols ? base[(size_t)(c + k) * col_stride + I] : 0;
        poseidon2_permutation(s);
    }
    if (last) {
        ulonglong2 *d = reinterpret_cast<ulonglong2 *>(digests + 4 * I);
        d[0] = make_ulonglong2(gl::canon(s[0]), gl::canon(s[1]));
        d[1] = make_ulonglong2(gl::canon(s[2]), gl::canon(s[3]));
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) capacity[(size_t)k * num_leaves + I] = s[8 + k];
    }
}

// leaf j = sponge( src0[jE..(j+1)E) || src1[jE..(j+1)E) || ... )           (merkle_tree.rs:176-386, FRI oracles)
__global__ void __launch_bounds__(256)
poseidon2_leaves_chunked_kernel(const u64 *src0, const u64 *src1, unsigned n_srcs, unsigned log_e, size_t num_leaves,
                                u64 *digests) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= num_leaves) return;
    const unsigned E = 1u << log_e;
    const unsigned total = n_srcs * E;
    u64 s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = 0;
    for (unsigned t = 0; t < total; t += 8) {   // one call site of the permutation (instruction-cache footprint, see above)
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const unsigned idx = t + k;             // wave-uniform
            u64 v = 0;
            if (idx < total) {
                const u64 *p = (idx >> log_e) == 0 ? src0 : src1;
                v = p[j * E + (idx & (E - 1))];
            }
            s[k] = v;
        }
        poseidon2_permutation(s);
    }
    ulonglong2 *d = reinterpret_cast<ulonglong2 *>(digests + 4 * j);
    d[0] = make_ulonglong2(gl::canon(s[0]), gl::canon(s[1]));
    d[1] = make_ulonglong2(gl::canon(s[2]), gl::canon(s[3]));
}

// node layer: parent i = perm(left || right || 0000)[0..4]                 (oracle/mod.rs:162-168)
__global__ void __launch_bounds__(256) poseidon2_nodes_kernel(const u64 *children, u64 *parents, size_t num_parents) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_parents) return;
    const ulonglong2 *c = reinterpret_cast<const ulonglong2 *>(children + 8 * i);
    ulonglong2 a = c[0], b = c[1], e = c[2], f = c[3];
    u64 s[12] = {a.x, a.y, b.x, b.y, e.x, e.y, f.x, f.y, 0, 0, 0, 0};
    poseidon2_permutation(s);
    ulonglong2 *d = reinterpret_cast<ulonglong2 *>(parents + 4 * i);
    d[0] = make_ulonglong2(gl::canon(s[0]), gl::canon(s[1]));
    d[1] = make_ulonglong2(gl::canon(s[2]), gl::canon(s[3]));
}

// ---------------------------------------------------------------------------------------------------------
// Small node layers: ONE permutation spread over 16 lanes (lane l holds state word l, lanes 12..15 idle).
// A permutation is a dependent chain of ~14 k instructions, so a layer with fewer nodes than the chip has lanes takes
// ~45 us however small it is — and the top ~13 layers of every tree (and every layer of the small FRI oracles) are such
// layers: 25-35 % of a 2^14..2^16-row proof and a fixed cost per proof that sharding over GPUs does not shrink.  Spread
// over lanes, the twelve S-boxes of a full round run side by side and both linear layers become the same routine — a
// 12-term dot product of the state (exchanged through LDS) with the lane's matrix row — which cuts the chain to ~4 k
// instructions.  Same arithmetic (weak residues, exact mod p), same digests.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 mad_vv(u32 x, u32 y, u64 c) {   // x*y + c, no carry out (callers bound the sums)
    u64 r;
    asm("v_mad_u64_u32 %[r], vcc, %[x], %[y], %[c]" : [r] "=v"(r) : [x] "v"(x), [y] "v"(y), [c] "v"(c) : "vcc");
    return r;
}
// sum_k coef[k] * x[k] for coefficients < 2^15: low and high words accumulate apart (each sum < 12 * 2^47 < 2^51)
__device__ __forceinline__ u64 lane_matvec(const u64 *x, const u32 (&coef)[12]) {
    u64 a0 = 0, a1 = 0, b0 = 0, b1 = 0;        // two chains per word to halve the dependent length
#pragma unroll
    for (int k = 0; k < 12; k += 2) {
        ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(x + k);
        a0 = mad_vv(gl::lo32(v.x), coef[k], a0);
        b0 = mad_vv(gl::hi32(v.x), coef[k], b0);
        a1 = mad_vv(gl::lo32(v.y), coef[k + 1], a1);
        b1 = mad_vv(gl::hi32(v.y), coef[k + 1], b1);
    }
    return combine_split(a0 + a1, b0 + b1);
}

constexpr int BJ_LANEPAR_GROUPS = 16;          // permutations per 256-lane workgroup
// this lane's rows of the external matrix circ(2*M4, M4, M4) and of the internal matrix 1 + diag(2^SH), and its round
// constants: a full round adds RC[12r + l] to word l, a partial round RC[12r] to word 0
struct LaneRows {
    u32 ext[12], inl[12];
    u64 rc[30];
    __device__ __forceinline__ void init(unsigned l) {
        constexpr u32 M4[4][4] = {{5, 7, 1, 3}, {4, 6, 1, 1}, {1, 3, 5, 7}, {1, 1, 4, 6}};
        constexpr u32 SH[12] = {4, 14, 11, 8, 0, 5, 2, 9, 13, 6, 3, 12};
#pragma unroll
        for (int k = 0; k < 12; k++) {
            ext[k] = l < 12 ? M4[l & 3][k & 3] * (((unsigned)k >> 2) == (l >> 2) ? 2u : 1u) : 0u;
            inl[k] = l < 12 ? 1u + ((unsigned)k == l ? (1u << SH[l < 12 ? l : 0]) : 0u) : 0u;
        }
#pragma unroll
        for (int r = 0; r < 30; r++) rc[r] = POSEIDON_RC[12 * r + ((r < 4 || r >= 26) && l < 12 ? l : 0)];
    }
};
// the permutation on a state spread over the lanes of group g (word l in lane l); every lane of the workgroup must call it
__device__ __forceinline__ u64 lanepar_permutation(u64 s, u64 (&xchg)[2][BJ_LANEPAR_GROUPS][16], unsigned g, unsigned l,
                                                   const LaneRows &rows) {
    xchg[0][g][l] = s;
    __syncthreads();
    s = lane_matvec(xchg[0][g], rows.ext);
#pragma unroll
    for (int r = 0; r < 30; r++) {
        const bool full = r < 4 || r >= 26;
        const u64 t = pow7w(addw_rc(s, rows.rc[r]));
        if (full || l == 0) s = t;
        xchg[(r + 1) & 1][g][l] = s;
        __syncthreads();
        s = full ? lane_matvec(xchg[(r + 1) & 1][g], rows.ext) : lane_matvec(xchg[(r + 1) & 1][g], rows.inl);
    }
    return s;
}

__global__ void __launch_bounds__(256)
poseidon2_nodes_lanepar_kernel(const u64 *children, u64 *parents, size_t num_parents) {
    __shared__ u64 xchg[2][BJ_LANEPAR_GROUPS][16];
    const unsigned g = threadIdx.x >> 4, l = threadIdx.x & 15;
    const size_t node = (size_t)blockIdx.x * BJ_LANEPAR_GROUPS + g;
    const bool live = node < num_parents;
    LaneRows rows;
    rows.init(l);
    u64 s = (live && l < 8) ? children[8 * node + l] : 0;
    s = lanepar_permutation(s, xchg, g, l, rows);
    if (live && l < 4) parents[4 * node + l] = gl::canon(s);
}

// the chunked leaf sponge of the small FRI oracles, one leaf per 16 lanes: lanes 0..7 overwrite the rate part with the
// next eight elements (zeros past the end), lanes 8..11 carry the capacity
__global__ void __launch_bounds__(256)
poseidon2_leaves_chunked_lanepar_kernel(const u64 *src0, const u64 *src1, unsigned n_srcs, unsigned log_e, size_t num_leaves,
                                        u64 *digests) {
    __shared__ u64 xchg[2][BJ_LANEPAR_GROUPS][16];
    const unsigned g = threadIdx.x >> 4, l = threadIdx.x & 15;
    const size_t j = (size_t)blockIdx.x * BJ_LANEPAR_GROUPS + g;
    const bool live = j < num_leaves;
    LaneRows rows;
    rows.init(l);
    const unsigned E = 1u << log_e, total = n_srcs * E;
    u64 s = 0;
    for (unsigned t = 0; t < total; t += 8) {
        if (l < 8) {
            const unsigned idx = t + l;
            u64 v = 0;
            if (live && idx < total) {
                const u64 *p = (idx >> log_e) == 0 ? src0 : src1;
                v = p[j * E + (idx & (E - 1))];
            }
            s = v;
        }
        s = lanepar_permutation(s, xchg, g, l, rows);
        __syncthreads();   // the exchange buffers are reused by the next block of the sponge
    }
    if (live && l < 4) digests[4 * j + l] = gl::canon(s);
}

__global__ void poseidon2_permute_states_kernel(u64 *states, size_t n_states) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_states) return;
    u64 s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = states[12 * i + k];
    poseidon2_permutation(s);
#pragma unroll
    for (int k = 0; k < 12; k++) states[12 * i + k] = gl::canon(s[k]);
}

static size_t nodes_lanepar_max() {   // layers up to this many parents use the lane-parallel kernel (BJ_NODES_LANEPAR_MAX=0: never)
    return bj::env().nodes_lanepar_max;
}
void launch_poseidon2_leaves(const u64 *d_base, size_t col_stride, const u64 *const *d_col_ptrs, unsigned n_cols,
                             size_t num_leaves, u64 *d_digests, hipStream_t s) {
    unsigned tpb = 256;
    hipLaunchKernelGGL(poseidon2_leaves_kernel, dim3((unsigned)((num_leaves + tpb - 1) / tpb)), dim3(tpb), 0, s,
                       d_base, col_stride, d_col_ptrs, n_cols, num_leaves, d_digests);
}
void launch_poseidon2_leaves_absorb(const u64 *d_base, size_t col_stride, unsigned n_cols, size_t num_leaves, u64 *d_capacity,
                                    u64 *d_digests, bool first, bool last, hipStream_t s) {
    hipLaunchKernelGGL(poseidon2_leaves_absorb_kernel, dim3((unsigned)((num_leaves + 255) / 256)), dim3(256), 0, s, d_base,
                       col_stride, n_cols, num_leaves, d_capacity, d_digests, first ? 1 : 0, last ? 1 : 0);
}
void launch_poseidon2_leaves_chunked(const u64 *d_src0, const u64 *d_src1, unsigned n_srcs, unsigned log_e,
                                     size_t num_leaves, u64 *d_digests, hipStream_t s) {
    unsigned tpb = 256;
    if (num_leaves <= nodes_lanepar_max())   // latency-bound: one leaf per 16 lanes
        hipLaunchKernelGGL(poseidon2_leaves_chunked_lanepar_kernel, dim3((unsigned)((num_leaves + BJ_LANEPAR_GROUPS - 1) / BJ_LANEPAR_GROUPS)),
                           dim3(tpb), 0, s, d_src0, d_src1, n_srcs, log_e, num_leaves, d_digests);
    else
        hipLaunchKernelGGL(poseidon2_leaves_chunked_kernel, dim3((unsigned)((num_leaves + tpb - 1) / tpb)), dim3(tpb), 0,
                           s, d_src0, d_src1, n_srcs, log_e, num_leaves, d_digests);
}
// tree layout: layer 0 = num_leaves digests, then num_leaves/2, ... down to cap_size (inclusive), back to back
void launch_poseidon2_node_layers(u64 *d_tree, size_t num_leaves, size_t cap_size, hipStream_t s) {
    u64 *prev = d_tree;
    size_t len = num_leaves;
    while (len > cap_size) {
        u64 *next = prev + 4 * len;
        size_t nl = len / 2;
        unsigned tpb = 256;
        if (nl <= nodes_lanepar_max())   // latency-bound layer: one permutation per 16 lanes
            hipLaunchKernelGGL(poseidon2_nodes_lanepar_kernel, dim3((unsigned)((nl + BJ_LANEPAR_GROUPS - 1) / BJ_LANEPAR_GROUPS)),
                               dim3(tpb), 0, s, prev, next, nl);
        else
            hipLaunchKernelGGL(poseidon2_nodes_kernel, dim3((unsigned)((nl + tpb - 1) / tpb)), dim3(tpb), 0, s, prev, next, nl);
        prev = next;
        len = nl;
    }
}
void launch_poseidon2_permute_states(u64 *d_states, size_t n_states, hipStream_t s) {
    unsigned tpb = 64;
    hipLaunchKernelGGL(poseidon2_permute_states_kernel, dim3((unsigned)((n_states + tpb - 1) / tpb)), dim3(tpb), 0, s,
                       d_states, n_states);
}

}  // namespace bj
