/* ORACLE — TEST INFRASTRUCTURE ONLY (see gl.h).
 * Goldilocks arithmetic of gl.h on eight lanes of an AVX-512 register (canonical residues in, canonical residues out), shared by
 * poseidon2_avx512.c and ntt_avx512.c.  The reference's own SIMD field is field/goldilocks/avx512_impl.rs (`MixedGL`: the same
 * mul_epu32 partial products and the reduction 2^64 = 2^32 - 1, 2^96 = -1 of goldilocks/mod.rs:188-201).  Every function carries its
 * own target attribute: the library is built for x86-64-v3 and these are called only after orc_poseidon2_avx512_available(). */
#ifndef ORACLE_GL_AVX512_H
#define ORACLE_GL_AVX512_H
#include "gl.h"
#include <immintrin.h>

#define AVX512 __attribute__((target("avx512f,avx512dq")))
typedef __m512i v8;

/* (lo, hi) of a 128-bit value per lane -> canonical residue: gl_reduce128 of gl.h, lane-wise */
AVX512 static inline v8 v8_reduce128(v8 lo, v8 hi) {
    const v8 eps = _mm512_set1_epi64((long long)GL_EPS), p = _mm512_set1_epi64((long long)GL_P);
    const v8 hi_hi = _mm512_srli_epi64(hi, 32), hi_lo = _mm512_and_si512(hi, eps);
    v8 t0 = _mm512_sub_epi64(lo, hi_hi);
    t0 = _mm512_mask_sub_epi64(t0, _mm512_cmplt_epu64_mask(lo, hi_hi), t0, eps);      /* borrow: subtract 2^64 mod p */
    const v8 t1 = _mm512_mul_epu32(hi_lo, eps);                                      /* hi_lo * (2^32 - 1) < 2^64 */
    v8 r = _mm512_add_epi64(t0, t1);
    r = _mm512_mask_add_epi64(r, _mm512_cmplt_epu64_mask(r, t1), r, eps);             /* carry */
    return _mm512_mask_sub_epi64(r, _mm512_cmpge_epu64_mask(r, p), r, p);
}
AVX512 static inline v8 v8_mul(v8 a, v8 b) {
    const v8 m32 = _mm512_set1_epi64(0xFFFFFFFFLL);
    const v8 ah = _mm512_srli_epi64(a, 32), bh = _mm512_srli_epi64(b, 32);
    const v8 ll = _mm512_mul_epu32(a, b), lh = _mm512_mul_epu32(a, bh), hl = _mm512_mul_epu32(ah, b), hh = _mm512_mul_epu32(ah, bh);
    const v8 mid = _mm512_add_epi64(lh, _mm512_srli_epi64(ll, 32));                   /* <= (2^32-1)^2 + 2^32 - 1: no wrap */
    const v8 mid2 = _mm512_add_epi64(hl, _mm512_and_si512(mid, m32));                 /* no wrap */
    const v8 lo = _mm512_or_si512(_mm512_and_si512(ll, m32), _mm512_slli_epi64(mid2, 32));
    const v8 hi = _mm512_add_epi64(hh, _mm512_add_epi64(_mm512_srli_epi64(mid, 32), _mm512_srli_epi64(mid2, 32)));
    return v8_reduce128(lo, hi);
}
AVX512 static inline v8 v8_add(v8 a, v8 b) {   /* gl_add */
    const v8 p = _mm512_set1_epi64((long long)GL_P);
    const v8 s = _mm512_add_epi64(a, b);
    const __mmask8 m = _mm512_cmplt_epu64_mask(s, a) | _mm512_cmpge_epu64_mask(s, p);
    return _mm512_mask_sub_epi64(s, m, s, p);
}
AVX512 static inline v8 v8_sub(v8 a, v8 b) {   /* gl_sub: a borrowed 2^64 = p + EPS gives EPS back */
    const v8 d = _mm512_sub_epi64(a, b);
    return _mm512_mask_sub_epi64(d, _mm512_cmplt_epu64_mask(a, b), d, _mm512_set1_epi64((long long)GL_EPS));
}
AVX512 static inline v8 v8_canon(v8 a) {
    const v8 p = _mm512_set1_epi64((long long)GL_P);
    return _mm512_mask_sub_epi64(a, _mm512_cmpge_epu64_mask(a, p), a, p);
}
#endif
