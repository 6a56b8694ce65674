/* ORACLE — TEST INFRASTRUCTURE ONLY (see gl.h).
 * The butterfly loops of ntt.c on AVX-512, eight butterflies per instruction: the same rounds, the same twiddles, the same canonical
 * residues as the scalar loops of serial_ct_ntt (fft/mod.rs:659-734) — a round whose pairs are at least eight apart pairs the
 * contiguous runs a[j .. j+8) and a[j+dist .. j+dist+8), so the vector form is the scalar loop with j stepping by eight.  The
 * reference's own CPU path does the same through `MixedGL` (field/goldilocks/avx512_impl.rs, fft/mod.rs:504-657).  ntt.c calls
 * these only when orc_poseidon2_avx512_available() (run-time CPU check; ORC_NO_AVX512=1 forces the scalar loops);
 * tests/test_oracle_avx512.py compares both paths, tests/test_oracle_ntt.py pins whichever runs against the naive DFT.
 */
#include "oracle.h"
#include "gl_avx512.h"

/* (u, v) <- (u + v*s, u - v*s) on count (a multiple of 8) pairs lo[j], hi[j]; canonical in and out */
AVX512 void orc_ct_bfly_x8(uint64_t *lo, uint64_t *hi, size_t count, uint64_t s) {
    const v8 vs = _mm512_set1_epi64((long long)s);
    for (size_t j = 0; j < count; j += 8) {
        const v8 u = _mm512_loadu_si512((const void *)(lo + j));
        const v8 v = v8_mul(_mm512_loadu_si512((const void *)(hi + j)), vs);
        _mm512_storeu_si512((void *)(hi + j), v8_sub(u, v));
        _mm512_storeu_si512((void *)(lo + j), v8_add(u, v));
    }
}
/* the first round (twiddle 1) */
AVX512 void orc_ct_addsub_x8(uint64_t *lo, uint64_t *hi, size_t count) {
    for (size_t j = 0; j < count; j += 8) {
        const v8 u = _mm512_loadu_si512((const void *)(lo + j)), v = _mm512_loadu_si512((const void *)(hi + j));
        _mm512_storeu_si512((void *)(hi + j), v8_sub(u, v));
        _mm512_storeu_si512((void *)(lo + j), v8_add(u, v));
    }
}
/* a[i] <- a[i] * s, count a multiple of 8 */
AVX512 void orc_scale_x8(uint64_t *a, size_t count, uint64_t s) {
    const v8 vs = _mm512_set1_epi64((long long)s);
    for (size_t j = 0; j < count; j += 8)
        _mm512_storeu_si512((void *)(a + j), v8_mul(_mm512_loadu_si512((const void *)(a + j)), vs));
}
/* distribute_powers (fft/mod.rs:308-317): a[i] <- a[i] * el^i, count a multiple of 8; lane l carries el^(8k + l) */
AVX512 void orc_distribute_powers_x8(uint64_t *a, size_t count, uint64_t el) {
    uint64_t p[8];
    p[0] = 1;
    for (int l = 1; l < 8; l++) p[l] = gl_mul(p[l - 1], el);
    const v8 step = _mm512_set1_epi64((long long)gl_mul(p[7], el));
    v8 pw = _mm512_loadu_si512((const void *)p);
    for (size_t j = 0; j < count; j += 8) {
        _mm512_storeu_si512((void *)(a + j), v8_mul(_mm512_loadu_si512((const void *)(a + j)), pw));
        pw = v8_mul(pw, step);
    }
}
AVX512 void orc_canonicalize_x8(uint64_t *a, size_t count) {
    for (size_t j = 0; j < count; j += 8) _mm512_storeu_si512((void *)(a + j), v8_canon(_mm512_loadu_si512((const void *)(a + j))));
}
