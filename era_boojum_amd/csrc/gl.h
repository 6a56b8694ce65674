// Goldilocks (p = 2^64 - 2^32 + 1) and F_p^2 = F_p[u]/(u^2 - 7) device arithmetic for gfx950.
//
// Semantics follow the reference field (src/field/goldilocks/mod.rs:188-255, 294-360; extension.rs:14-16;
// field/traits/field.rs:407-512) as functions mod p.  The reference lets in-memory values be any u64 and reduces
// on compare/serialise; here every value is canonicalised when it is loaded from HBM (gl_load) and all arithmetic
// keeps values in [0, p), so what is stored back is always the canonical residue — parity with the reference is
// equality of canonical residues.
//
// CDNA4 has no 64x64 multiplier: a field multiplication is four v_mad_u64_u32 (quarter rate) plus a shift/add
// reduction using 2^64 = 2^32 - 1 and 2^96 = -1 (mod p).
#pragma once
#if !defined(__HIPCC_RTC__)   // hiprtc (run-time compiled gate kernels, gate_jit.hip) brings the runtime declarations itself
#include <hip/hip_runtime.h>
#include <stdint.h>
#else
typedef unsigned int uint32_t;
typedef unsigned long uint64_t;
#endif

namespace gl {

typedef uint64_t u64;
typedef uint32_t u32;

static constexpr u64 P = 0xFFFFFFFF00000001ULL;
static constexpr u64 EPS = 0xFFFFFFFFULL;  // 2^64 mod p
static constexpr u64 GEN = 7;              // multiplicative generator = LDE coset shift = F_p^2 non-residue

__host__ __device__ __forceinline__ u64 canon(u64 a) { return a >= P ? a - P : a; }

// 32-bit limb helpers.  The device code is written as explicit 32-bit carry chains (v_add_co/v_addc, v_sub_co/v_subb):
// on gfx950 a 32-bit VALU op issues in ~2 cycles per wave64, a 64-bit one (v_lshl_add_u64, v_cmp_*_u64) in ~4 and
// v_mad_u64_u32 in ~5, so carry chains beat "64-bit add + 64-bit compare + select" by a wide margin
// (tools/microbench.hip: field mul 90 -> ~60 cycles, add/sub 26 -> ~14).
__host__ __device__ __forceinline__ u32 lo32(u64 a) { return (u32)a; }
__host__ __device__ __forceinline__ u32 hi32(u64 a) { return (u32)(a >> 32); }
__host__ __device__ __forceinline__ u64 pack(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }

// a, b in [0,p) -> (a + b) mod p
__host__ __device__ __forceinline__ u64 add(u64 a, u64 b) {
#if !defined(__HIP_DEVICE_COMPILE__)
    u64 s = a + b;                                    // host: 64-bit words
    if (s < a) return s + 0xFFFFFFFFULL;              // wrapped: a + b - 2^64 + EPS = a + b - p
    return s >= P ? s - P : s;
#endif
    u32 c1, c2, c3, c4;
    u32 s0 = __builtin_addc(lo32(a), lo32(b), 0u, &c1);
    u32 s1 = __builtin_addc(hi32(a), hi32(b), c1, &c2);
    u32 t0 = __builtin_addc(s0, 0xFFFFFFFFu, 0u, &c3);  // s + EPS = s - p (mod 2^64)
    u32 t1 = __builtin_addc(s1, 0u, c3, &c4);
    // a + b wrapped (c2)  or  s >= p (s + EPS wraps, c4)  ->  take s - p
    return (c2 | c4) ? pack(t0, t1) : pack(s0, s1);
}
// a, b in [0,p) -> (a - b) mod p
__host__ __device__ __forceinline__ u64 sub(u64 a, u64 b) {
#if !defined(__HIP_DEVICE_COMPILE__)
    return a >= b ? a - b : a - b - 0xFFFFFFFFULL;   // host: borrowed 2^64 = p + EPS
#endif
    u32 b1, b2, b3, b4;
    u32 d0 = __builtin_subc(lo32(a), lo32(b), 0u, &b1);
    u32 d1 = __builtin_subc(hi32(a), hi32(b), b1, &b2);
    u32 e = b2 ? 0xFFFFFFFFu : 0u;  // + p == - EPS (mod 2^64)
    d0 = __builtin_subc(d0, e, 0u, &b3);
    d1 = __builtin_subc(d1, 0u, b3, &b4);
    return pack(d0, d1);
}
__host__ __device__ __forceinline__ u64 neg(u64 a) { return a ? P - a : 0; }
__host__ __device__ __forceinline__ u64 dbl(u64 a) { return add(a, a); }

// (hi_hi : hi_lo : lo) = 128-bit value -> canonical residue, using 2^64 = 2^32 - 1 and 2^96 = -1 (mod p)
__host__ __device__ __forceinline__ u64 reduce_limbs(u32 hi_hi, u32 hi_lo, u64 lo) {
#if !defined(__HIP_DEVICE_COMPILE__)
    // host: the same identities on 64-bit words (the limb chains below are written for the GPU's 32-bit VALU)
    u64 t = lo - hi_hi;
    if (lo < hi_hi) t -= 0xFFFFFFFFULL;              // borrowed 2^64 = p + EPS: add p back
    const u64 m = (u64)hi_lo * 0xFFFFFFFFULL;
    u64 r = t + m;
    if (r < m) r += 0xFFFFFFFFULL;                   // wrapped: 2^64 = EPS (mod p); r < 2^64 - 2^33 then, no second wrap
    return r >= P ? r - P : r;
#else
    u32 b1, b2, b3, b4;
    // t0 = lo - hi_hi  (+p on borrow), in [0, 2^64)
    u32 d0 = __builtin_subc(lo32(lo), hi_hi, 0u, &b1);
    u32 d1 = __builtin_subc(hi32(lo), 0u, b1, &b2);
    u32 e = b2 ? 0xFFFFFFFFu : 0u;
    d0 = __builtin_subc(d0, e, 0u, &b3);
    d1 = __builtin_subc(d1, 0u, b3, &b4);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BJ_GL_REDUCE_CHAINS)
    // r = t0 + hi_lo * EPS in ONE multiply-add; "+EPS" when the sum wrapped (then r < 2^64 - 2^33: no second wrap, result
    // < p) or when r >= p (r - p = r + EPS mod 2^64).  Carry-writing VALU ops are the scarce resource on gfx950 (one per
    // ~4.4 cycles per SIMD), this form needs 2 instead of 6 of them.  gfx950 needs 2 wait states between a VALU write of
    // an SGPR/VCC and a VALU read of it as a mask; the compiler cannot see inside the string, so they are placed by
    // hand (v_cmp + s_nop 0 after the mad, s_nop 0 + v_cndmask after the v_cmp).  No SALU combination of the masks:
    // an s_or_b64 right after the VALU writes read stale values on a lone wave.
    u64 r, cm;
    u32 fix;
    asm("v_mad_u64_u32 %[r], %[cm], %[m], -1, %[t0]\n\t"
        "v_cmp_ge_u64 vcc, %[r], %[p]\n\t"
        "s_nop 0\n\t"
        "v_cndmask_b32 %[fix], 0, -1, %[cm]\n\t"
        "v_cndmask_b32 %[fix], %[fix], -1, vcc"
        : [r] "=&v"(r), [cm] "=&s"(cm), [fix] "=&v"(fix)
        : [m] "v"(hi_lo), [t0] "v"(pack(d0, d1)), [p] "s"(P)
        : "vcc");
    return r + (u64)fix;
#else
    u32 b5, b6, c1, c2, c3, c4;
    // t1 = hi_lo * (2^32 - 1) = (hi_lo << 32) - hi_lo
    u32 m0 = __builtin_subc(0u, hi_lo, 0u, &b5);
    u32 m1 = __builtin_subc(hi_lo, 0u, b5, &b6);
    // r = t0 + t1, then the carry fix and the canonicalisation share one "+ EPS"
    u32 r0 = __builtin_addc(d0, m0, 0u, &c1);
    u32 r1 = __builtin_addc(d1, m1, c1, &c2);
    u32 q0 = __builtin_addc(r0, 0xFFFFFFFFu, 0u, &c3);
    u32 q1 = __builtin_addc(r1, 0u, c3, &c4);
    return (c2 | c4) ? pack(q0, q1) : pack(r0, r1);
#endif
#endif
}
__host__ __device__ __forceinline__ u64 reduce128(u64 hi, u64 lo) { return reduce_limbs(hi32(hi), lo32(hi), lo); }

// 64x64 -> (hi_hi, hi_lo, lo) limbs of the 128-bit product.
// Device: four v_mad_u64_u32 issued from one inline-asm block so that the cross-term accumulation keeps its
// carry-out (the C++ formulation costs five extra v_mov to build {x,0} addend pairs).  The two independent mads
// between the carry-producing mad and the v_cndmask that reads it cover the 2 wait states gfx950 requires between a
// VALU SGPR write and a VALU read of that SGPR (the compiler cannot see inside the asm string).
__host__ __device__ __forceinline__ void mul_limbs(u64 a, u64 b, u32 &hi_hi, u32 &hi_lo, u64 &lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    u32 a0 = lo32(a), a1 = hi32(a), b0 = lo32(b), b1 = hi32(b);
    u64 L, X, H, cmask;
    u32 cx;
    asm("v_mad_u64_u32 %[X], vcc, %[a0], %[b1], 0\n\t"
        "v_mad_u64_u32 %[X], %[cm], %[a1], %[b0], %[X]\n\t"
        "v_mad_u64_u32 %[L], vcc, %[a0], %[b0], 0\n\t"
        "v_mad_u64_u32 %[H], vcc, %[a1], %[b1], 0\n\t"
        "v_cndmask_b32 %[cx], 0, 1, %[cm]"
        : [L] "=&v"(L), [X] "=&v"(X), [H] "=&v"(H), [cx] "=v"(cx), [cm] "=&s"(cmask)
        : [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1)
        : "vcc");
    // product = L + X*2^32 + cx*2^96 + H*2^64
    u32 c1, c2, c3;
    u32 lo_hi = __builtin_addc(hi32(L), lo32(X), 0u, &c1);
    hi_lo = __builtin_addc(lo32(H), hi32(X), c1, &c2);
    hi_hi = __builtin_addc(hi32(H), cx, c2, &c3);  // cannot overflow: the product is < 2^128
    lo = pack(lo32(L), lo_hi);
#else
    unsigned __int128 x = (unsigned __int128)a * b;
    u64 hi = (u64)(x >> 64);
    hi_hi = hi32(hi);
    hi_lo = lo32(hi);
    lo = (u64)x;
#endif
}

#if defined(__HIP_DEVICE_COMPILE__)
// any u64 x any u64 -> "weak" residue (some u64 congruent to a*b mod p, not necessarily < p), 8 half-rate + 4 full-rate VALU
// instructions (the limb-assembly + canonical reduction above is 14 + 3 and needs six carry links):
//   T = a0*b0;  X = a1*b0 + (a0*b1 + T.hi)  [carry cm, weight 2^96];  H = a1*b1 + X.hi       -- every addition rides on a
//   multiply-add, so   a*b = (T.lo | X.lo << 32) + H * 2^64 + cm * 2^96   exactly, with H < 2^64;
//   2^64 = EPS, 2^96 = -1 (mod p):   a*b == lo + H.lo * EPS - (H.hi + cm)
//   R = lo + H.lo*EPS mod 2^64 (carry c, ONE multiply-add);  D = R - H.hi - cm mod 2^64 (borrow b; cm enters as the borrow-in)
//   result = D + c*EPS mod 2^64 (one more multiply-add on the 0/1 carry).  Why that is right:
//     b = 0: c = 1 means R < 2^64 - 2^33 + 1, so D + EPS cannot wrap;
//     b = 1: D >= 2^64 - 2^32 + 1 = p (H.hi + cm <= 2^32 - 1); with c = 1 the wrap of D + EPS cancels the borrow;
//            with c = 0 the value is D - EPS (no second borrow as D >= p).  That last case needs R < 2^32: probability
//            2^-32 per product, so it is a wave-uniform branch around three instructions, not predicated code.
// Temporaries are fixed physical registers (clobbers): the sequence writes halves of 64-bit pairs, which operands allocated by
// the compiler cannot express (no sub-register modifier for inline-asm operands on this target).  The s_nop covers the 2 wait
// states gfx950 needs between a VALU write of an SGPR pair and a VALU read of it (the compiler cannot see inside the string).
__device__ __forceinline__ u64 mul_weak(u64 a, u64 b) {
    u64 out, cm, c;
    asm("v_mad_u64_u32 v[48:49], vcc, %[a0], %[b0], 0\n\t"
        "v_mov_b32 v51, 0\n\t"
        "v_mov_b32 v50, v49\n\t"
        "v_mad_u64_u32 v[50:51], vcc, %[a0], %[b1], v[50:51]\n\t"
        "v_mad_u64_u32 v[50:51], %[cm], %[a1], %[b0], v[50:51]\n\t"
        "v_mov_b32 v53, 0\n\t"
        "v_mov_b32 v52, v51\n\t"
        "v_mad_u64_u32 v[52:53], vcc, %[a1], %[b1], v[52:53]\n\t"
        "v_mov_b32 v49, v50\n\t"
        "v_mad_u64_u32 v[50:51], %[c], v52, -1, v[48:49]\n\t"
        "v_subb_co_u32 v50, vcc, v50, v53, %[cm]\n\t"
        "s_nop 0\n\t"
        "v_cndmask_b32 v52, 0, 1, %[c]\n\t"
        "v_subb_co_u32 v51, vcc, v51, 0, vcc\n\t"
        "s_cbranch_vccz .Lglmw%=\n\t"
        "s_andn2_b64 vcc, vcc, %[c]\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32 v53, 0, -1, vcc\n\t"
        "v_sub_co_u32 v50, vcc, v50, v53\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 v51, vcc, 0, v51, vcc\n"
        ".Lglmw%=:\n\t"
        "v_mad_u64_u32 %[out], vcc, v52, -1, v[50:51]"
        : [out] "=&v"(out), [cm] "=&s"(cm), [c] "=&s"(c)
        : [a0] "v"(lo32(a)), [a1] "v"(hi32(a)), [b0] "v"(lo32(b)), [b1] "v"(hi32(b))
        : "vcc", "scc", "v48", "v49", "v50", "v51", "v52", "v53");   // scc: the s_andn2_b64 of the borrow path writes it
    return out;
}
#else
__host__ __forceinline__ u64 mul_weak(u64 a, u64 b) {   // host pass of the kernels' translation units: any representative will do
    u32 hi_hi, hi_lo;
    u64 lo;
    mul_limbs(a, b, hi_hi, hi_lo, lo);
    return reduce_limbs(hi_hi, hi_lo, lo);
}
#endif

__host__ __device__ __forceinline__ u64 mul(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BJ_GL_MUL_LIMBS)
    return canon(mul_weak(a, b));
#else
    u32 hi_hi, hi_lo;
    u64 lo;
    mul_limbs(a, b, hi_hi, hi_lo, lo);
    return reduce_limbs(hi_hi, hi_lo, lo);
#endif
}
__host__ __device__ __forceinline__ u64 sqr(u64 a) { return mul(a, a); }

// Sums and differences of WEAK residues (any u64 in, any u64 congruent to the result out): what the product chains of the
// copy-permutation quotient run between their weak products, so that nothing is canonicalised inside a 184-product chain.
//   a + b wraps 2^64 -> + EPS (= 2^64 mod p); that can wrap once more only when both operands are within 2^32 of 2^64, and
//   then the result is below 2^32 and takes the second EPS without a third wrap.  Same with borrows for a - b.
__host__ __device__ __forceinline__ u64 add_weak(u64 a, u64 b) {
    u32 c1, c2, c3, c4;
    u32 s0 = __builtin_addc(lo32(a), lo32(b), 0u, &c1);
    u32 s1 = __builtin_addc(hi32(a), hi32(b), c1, &c2);
    s0 = __builtin_addc(s0, c2 ? 0xFFFFFFFFu : 0u, 0u, &c3);
    s1 = __builtin_addc(s1, 0u, c3, &c4);
    const u64 r = pack(s0, s1);
    return c4 ? r + EPS : r;
}
__host__ __device__ __forceinline__ u64 sub_weak(u64 a, u64 b) {
    u32 b1, b2, b3, b4;
    u32 d0 = __builtin_subc(lo32(a), lo32(b), 0u, &b1);
    u32 d1 = __builtin_subc(hi32(a), hi32(b), b1, &b2);
    d0 = __builtin_subc(d0, b2 ? 0xFFFFFFFFu : 0u, 0u, &b3);   // borrowed 2^64 = p + EPS: take EPS off again
    d1 = __builtin_subc(d1, 0u, b3, &b4);
    const u64 r = pack(d0, d1);
    return b4 ? r - EPS : r;                                    // second borrow: the value was within 2^32 of 0 from below
}
// 7 * a on weak residues (the non-residue of F_p^2): the 67-bit product folded with 2^64 = EPS — three multiply-adds and a
// carry fix instead of a full product
__host__ __device__ __forceinline__ u64 mul7_weak(u64 a) {
    const u64 lo = (u64)lo32(a) * 7u;
    const u64 hi = (u64)hi32(a) * 7u + hi32(lo);       // 7 a = lo32(lo) + hi * 2^32,  hi < 7 * 2^32
    const u64 t = (u64)hi32(hi) * EPS;                 // the bits above 2^64 (at most 6), times 2^64 mod p
    const u64 s = pack(lo32(lo), lo32(hi)) + t;
    return s < t ? s + EPS : s;                        // wrapped: s < 2^35 now, the second EPS cannot wrap
}

// any u64 times a 32-bit integer -> weak residue: two 32 x 32 products, the (at most 32) bits above 2^64 folded back with one
// multiply-add by 2^64 mod p — 3 multiply-adds and a conditional correction instead of the 15 instructions of mul_weak.  For the
// copy-permutation non-residues k_c (small integers: 1, 7, 11, 13, ... by make_non_residues, utils.rs:636-688).
__host__ __device__ __forceinline__ u64 mul_u32_weak(u64 a, u32 k) {
    const u64 lo = (u64)lo32(a) * k;
    const u64 hi = (u64)hi32(a) * k + hi32(lo);        // k a = lo32(lo) + hi * 2^32,  hi <= (2^32 - 1)^2 + 2^32 - 1 < 2^64
    const u64 t = (u64)hi32(hi) * EPS;                 // the bits above 2^64, times 2^64 mod p;  t <= (2^32 - 1)^2
    const u64 s = pack(lo32(lo), lo32(hi)) + t;
    return s < t ? s + EPS : s;                        // wrapped: s < t <= 2^64 - 2^33 + 1, so s + EPS < 2^64: no second wrap
}

}  // namespace gl
#include "gl_asm.inc"   // generated: butterfly2_weak_asm, addsub2_weak_asm (tools/gen_gl_asm.py)
namespace gl {
#if defined(__HIP_DEVICE_COMPILE__)
// Lazy radix-2 butterflies of the NTT kernels, two at a time (the two chains fill each other's SGPR wait states):
//   (u, v) <- (u + v * w, u - v * w)  on weak residues, 15 half-rate + 6 full-rate VALU instructions per butterfly instead of
//   the 21 + 9 of canonical mul / add / sub.  The wave-uniform fallback (a second wrap of the sum or the difference, the
//   borrow-without-carry product: ~2^-32 per butterfly) recomputes all four results canonically.
struct bfly4 {
    u64 sa, da, sb, db;
};
// the slow path is ONE out-of-line function per kernel image: inlined it is ~100 instructions at each of the 96 butterfly
// sites of a three-step NTT kernel, more code than the instruction cache holds
__device__ __attribute__((noinline)) bfly4 butterfly2_canonical(u64 ua, u64 va, u64 wa, u64 ub, u64 vb, u64 wb) {
    const u64 ta = mul(va, wa), tb = mul(vb, wb), ca = canon(ua), cb = canon(ub);
    return {add(ca, ta), sub(ca, ta), add(cb, tb), sub(cb, tb)};
}
__device__ __forceinline__ void butterfly2_weak(u64 &ua, u64 &va, u64 wa, u64 &ub, u64 &vb, u64 wb) {
    u64 sa, da, sb, db;
    const u64 rare = butterfly2_weak_asm(ua, va, wa, ub, vb, wb, sa, da, sb, db);
    if (__builtin_expect(rare != 0, 0)) {
        const bfly4 r = butterfly2_canonical(ua, va, wa, ub, vb, wb);
        sa = r.sa; da = r.da; sb = r.sb; db = r.db;
    }
    ua = sa; va = da; ub = sb; vb = db;
}
__device__ __forceinline__ void addsub2_weak(u64 &ua, u64 &va, u64 &ub, u64 &vb) {
    u64 sa, da, sb, db;
    const u64 rare = addsub2_weak_asm(ua, va, ub, vb, sa, da, sb, db);
    if (__builtin_expect(rare != 0, 0)) {
        const bfly4 r = butterfly2_canonical(ua, va, 1, ub, vb, 1);
        sa = r.sa; da = r.da; sb = r.sb; db = r.db;
    }
    ua = sa; va = da; ub = sb; vb = db;
}
#else
__host__ __forceinline__ void butterfly2_weak(u64 &ua, u64 &va, u64 wa, u64 &ub, u64 &vb, u64 wb) {   // host pass of the kernels' TUs
    const u64 ta = mul(canon(va), canon(wa)), tb = mul(canon(vb), canon(wb)), ca = canon(ua), cb = canon(ub);
    ua = add(ca, ta); va = sub(ca, ta); ub = add(cb, tb); vb = sub(cb, tb);
}
__host__ __forceinline__ void addsub2_weak(u64 &ua, u64 &va, u64 &ub, u64 &vb) {
    const u64 ca = canon(ua), cb = canon(ub), ta = canon(va), tb = canon(vb);
    ua = add(ca, ta); va = sub(ca, ta); ub = add(cb, tb); vb = sub(cb, tb);
}
#endif

// a * 2^k mod p for 0 <= k < 32 (shift instead of multiply; used by Poseidon2's internal matrix): the part shifted
// out is < 2^32, so the reduction is  lo + out*(2^32-1)  with one wrap/canonicalisation fix
__host__ __device__ __forceinline__ u64 mul_pow2(u64 a, unsigned k) {
    if (k == 0) return a;
    u32 out = (u32)(a >> (64 - k));
    u64 lo = a << k;
    u32 b5, b6, c1, c2, c3, c4;
    u32 m0 = __builtin_subc(0u, out, 0u, &b5);
    u32 m1 = __builtin_subc(out, 0u, b5, &b6);
    u32 r0 = __builtin_addc(lo32(lo), m0, 0u, &c1);
    u32 r1 = __builtin_addc(hi32(lo), m1, c1, &c2);
    u32 q0 = __builtin_addc(r0, 0xFFFFFFFFu, 0u, &c3);
    u32 q1 = __builtin_addc(r1, 0u, c3, &c4);
    return (c2 | c4) ? pack(q0, q1) : pack(r0, r1);
}

__host__ __device__ inline u64 pow(u64 a, u64 e) {
    u64 r = 1;
    while (e) {
        if (e & 1) r = mul(r, a);
        a = sqr(a);
        e >>= 1;
    }
    return r;
}
__host__ __device__ inline u64 inv(u64 a) { return pow(a, P - 2); }

// domain_generator_for_size (cs/implementations/utils.rs:13-28); radix_2_subgroup_generator = 0x185629dcda58878c
__host__ __device__ inline u64 omega(unsigned log_n) {
    u64 w = 0x185629dcda58878cULL;
    for (unsigned i = log_n; i < 32; i++) w = sqr(w);
    return w;
}

// ---- quadratic extension, stored as two base columns (c0, c1), never interleaved ----
struct e2 {
    u64 c0, c1;
};
__host__ __device__ __forceinline__ e2 e2_add(e2 a, e2 b) { return {add(a.c0, b.c0), add(a.c1, b.c1)}; }
__host__ __device__ __forceinline__ e2 e2_sub(e2 a, e2 b) { return {sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
__host__ __device__ __forceinline__ e2 e2_mul(e2 a, e2 b) {  // Karatsuba, field.rs:407-426
    u64 v0 = mul(a.c0, b.c0), v1 = mul(a.c1, b.c1);
    u64 c1 = sub(sub(mul(add(a.c0, a.c1), add(b.c0, b.c1)), v0), v1);
    u64 seven_v1 = sub(mul_pow2(v1, 3), v1);
    return {add(v0, seven_v1), c1};
}
// the same product on weak residues, schoolbook: four weak products, two weak sums, 7 * a1 b1 through mul7_weak — no
// canonicalisation anywhere (Karatsuba's three products cost five canonical additions / subtractions on this VALU)
__host__ __device__ __forceinline__ e2 e2_mul_weak(e2 a, e2 b) {
    const u64 v1 = mul_weak(a.c1, b.c1);
    return {add_weak(mul_weak(a.c0, b.c0), mul7_weak(v1)), add_weak(mul_weak(a.c0, b.c1), mul_weak(a.c1, b.c0))};
}
__host__ __device__ __forceinline__ e2 e2_sqr(e2 a) { return e2_mul(a, a); }
__host__ __device__ __forceinline__ e2 e2_mul_base(e2 a, u64 s) { return {mul(a.c0, s), mul(a.c1, s)}; }
__host__ __device__ inline e2 e2_inv(e2 a) {  // field.rs:484-512
    u64 n = sub(sqr(a.c0), mul(GEN, sqr(a.c1)));
    u64 ni = inv(n);
    return {mul(a.c0, ni), neg(mul(a.c1, ni))};
}

__host__ __device__ __forceinline__ u32 bitrev32(u32 x, unsigned bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return bits ? (__brev(x) >> (32 - bits)) : 0;
#else
    u32 r = 0;
    for (unsigned i = 0; i < bits; i++) {
        r = (r << 1) | (x & 1);
        x >>= 1;
    }
    return r;
#endif
}

}  // namespace gl
