/* ORACLE — TEST INFRASTRUCTURE ONLY (see gl.h).
 * The Poseidon2 permutation of poseidon2.c on AVX-512: EIGHT independent permutations side by side, lane l of every zmm register
 * holding state word i of permutation l.  The reference's own SIMD path (implementations/poseidon2/state_avx512.rs, 684 lines;
 * field arithmetic field/goldilocks/avx512_impl.rs `MixedGL`) spreads the twelve words of ONE state over two registers; a checker
 * that hashes 2^26 independent leaves has eight leaves to fill the lanes with instead, and then the algorithm is literally the
 * scalar one of state_generic_impl.rs:128-233 executed lane-wise — same rounds, same constants, same matrices:
 *   x^7                      state_generic_impl.rs:142-149      (4 products)
 *   external matrix          suggested_mds.rs:21-103            circ(2 M4, M4, M4), on the 32-bit halves of the words (sums < 2^40)
 *   internal matrix          params.rs:38-39, state_generic_impl.rs:171-219   1 + diag(2^k), on the halves (< 2^47)
 *   product and reduction    goldilocks/mod.rs:188-201, avx512_impl.rs (mul_epu32 partial products, 2^64 = 2^32 - 1, 2^96 = -1)
 * It is pinned by the same vectors as the scalar code: tests/test_oracle_fixture.py runs the golden proof's leaves, nodes and
 * paths through both, tests/test_oracle_avx512.py compares the two on random and edge states.  Selected at run time
 * (__builtin_cpu_supports: the library is built for x86-64-v3 and travels to other hosts); ORC_NO_AVX512=1 forces the scalar code.
 */
#include "oracle.h"
#include "poseidon_rc.h"
#include "gl_avx512.h"
#include <stdlib.h>
#include <string.h>

static const uint64_t RC[BJ_POSEIDON_NUM_RC] = BJ_POSEIDON_RC_TABLE;
static const unsigned SH[12] = {4, 14, 11, 8, 0, 5, 2, 9, 13, 6, 3, 12};

int orc_poseidon2_avx512_available(void) {
    static int cached = -1;
    if (cached < 0) {
        const char *off = getenv("ORC_NO_AVX512");
        __builtin_cpu_init();
        cached = (!off || !*off || *off == '0') && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq");
    }
    return cached;
}

#define reduce128 v8_reduce128
#define mul v8_mul
#define add v8_add
AVX512 static inline v8 pow7(v8 x) {
    const v8 x2 = mul(x, x), x3 = mul(x2, x), x4 = mul(x2, x2);
    return mul(x4, x3);
}
/* a + b * 2^32 with a, b < 2^50 (the sums of the low and of the high halves of a linear layer) -> canonical residue */
AVX512 static inline v8 fold_halves(v8 a, v8 b) {
    const v8 bl = _mm512_slli_epi64(b, 32);
    const v8 lo = _mm512_add_epi64(a, bl);
    const v8 hi = _mm512_mask_add_epi64(_mm512_srli_epi64(b, 32), _mm512_cmplt_epu64_mask(lo, bl), _mm512_srli_epi64(b, 32),
                                        _mm512_set1_epi64(1));
    return reduce128(lo, hi);
}
/* suggested_mds.rs:21-56 on one plane of halves (every value < 2^32 in, < 2^36 out) */
AVX512 static inline void block_mul(const v8 *x, v8 *y) {
    const v8 t0 = _mm512_add_epi64(x[0], x[1]), t1 = _mm512_add_epi64(x[2], x[3]);
    const v8 t2 = _mm512_add_epi64(_mm512_slli_epi64(x[1], 1), t1), t3 = _mm512_add_epi64(_mm512_slli_epi64(x[3], 1), t0);
    const v8 t4 = _mm512_add_epi64(_mm512_slli_epi64(t1, 2), t3), t5 = _mm512_add_epi64(_mm512_slli_epi64(t0, 2), t2);
    y[0] = _mm512_add_epi64(t3, t5); y[1] = t5; y[2] = _mm512_add_epi64(t2, t4); y[3] = t4;
}
AVX512 static inline void ext_mds(v8 *s) {       /* suggested_mds.rs:59-103 */
    const v8 m32 = _mm512_set1_epi64(0xFFFFFFFFLL);
    v8 lo[12], hi[12], yl[12], yh[12];
    for (int i = 0; i < 12; i++) { lo[i] = _mm512_and_si512(s[i], m32); hi[i] = _mm512_srli_epi64(s[i], 32); }
    for (int b = 0; b < 3; b++) { block_mul(lo + 4 * b, yl + 4 * b); block_mul(hi + 4 * b, yh + 4 * b); }
    for (int j = 0; j < 4; j++) {
        const v8 sl = _mm512_add_epi64(yl[j], _mm512_add_epi64(yl[4 + j], yl[8 + j]));
        const v8 sh = _mm512_add_epi64(yh[j], _mm512_add_epi64(yh[4 + j], yh[8 + j]));
        for (int b = 0; b < 3; b++)
            s[4 * b + j] = fold_halves(_mm512_add_epi64(yl[4 * b + j], sl), _mm512_add_epi64(yh[4 * b + j], sh));
    }
}
AVX512 static inline void full_round(v8 *s, int r) {      /* state_generic_impl.rs:158-168 */
    for (int i = 0; i < 12; i++) s[i] = pow7(add(s[i], _mm512_set1_epi64((long long)gl_canon(RC[12 * r + i]))));
    ext_mds(s);
}
AVX512 static inline void partial_round(v8 *s, int r) {   /* state_generic_impl.rs:171-219 */
    const v8 m32 = _mm512_set1_epi64(0xFFFFFFFFLL);
    s[0] = pow7(add(s[0], _mm512_set1_epi64((long long)gl_canon(RC[12 * r]))));
    v8 lo[12], hi[12], sl = _mm512_setzero_si512(), sh = _mm512_setzero_si512();
    for (int i = 0; i < 12; i++) {
        lo[i] = _mm512_and_si512(s[i], m32); hi[i] = _mm512_srli_epi64(s[i], 32);
        sl = _mm512_add_epi64(sl, lo[i]); sh = _mm512_add_epi64(sh, hi[i]);
    }
    for (int i = 0; i < 12; i++)
        s[i] = fold_halves(_mm512_add_epi64(_mm512_slli_epi64(lo[i], SH[i]), sl), _mm512_add_epi64(_mm512_slli_epi64(hi[i], SH[i]), sh));
}
/* the permutation on eight canonical states, word-major: s[i] = word i of the eight */
AVX512 static inline void permute8(v8 *s) {                /* state_generic_impl.rs:221-233 */
    ext_mds(s);
    int r = 0;
    for (int i = 0; i < 4; i++) full_round(s, r++);
    for (int i = 0; i < 22; i++) partial_round(s, r++);
    for (int i = 0; i < 4; i++) full_round(s, r++);
}
#define canon8 v8_canon
/* TWO groups of eight states: a partial round raises ONE word per state to the seventh power — four products that depend on each
 * other, one register wide — so a single group leaves the multiplier idle for most of the 22 partial rounds; two groups side by
 * side give the scheduler a second chain to fill the latency with.  Same permutation on each group. */
AVX512 static inline void partial_round2(v8 *a, v8 *b, int r) {
    const v8 m32 = _mm512_set1_epi64(0xFFFFFFFFLL), rc = _mm512_set1_epi64((long long)gl_canon(RC[12 * r]));
    const v8 xa = add(a[0], rc), xb = add(b[0], rc);
    const v8 xa2 = mul(xa, xa), xb2 = mul(xb, xb);
    const v8 xa3 = mul(xa2, xa), xb3 = mul(xb2, xb);
    const v8 xa4 = mul(xa2, xa2), xb4 = mul(xb2, xb2);
    a[0] = mul(xa4, xa3);
    b[0] = mul(xb4, xb3);
    for (int g = 0; g < 2; g++) {
        v8 *s = g ? b : a;
        v8 lo[12], hi[12], sl = _mm512_setzero_si512(), sh = _mm512_setzero_si512();
        for (int i = 0; i < 12; i++) {
            lo[i] = _mm512_and_si512(s[i], m32); hi[i] = _mm512_srli_epi64(s[i], 32);
            sl = _mm512_add_epi64(sl, lo[i]); sh = _mm512_add_epi64(sh, hi[i]);
        }
        for (int i = 0; i < 12; i++)
            s[i] = fold_halves(_mm512_add_epi64(_mm512_slli_epi64(lo[i], SH[i]), sl), _mm512_add_epi64(_mm512_slli_epi64(hi[i], SH[i]), sh));
    }
}
AVX512 static inline void permute8x2(v8 *a, v8 *b) {
    ext_mds(a); ext_mds(b);
    int r = 0;
    for (int i = 0; i < 4; i++, r++) { full_round(a, r); full_round(b, r); }
    for (int i = 0; i < 22; i++) partial_round2(a, b, r++);
    for (int i = 0; i < 4; i++, r++) { full_round(a, r); full_round(b, r); }
}

/* eight states back to back (8 x 12 words, state-major as orc_poseidon2_permutation takes one): for the tests */
AVX512 void orc_poseidon2_permutation_x8(uint64_t *states) {
    v8 s[12];
    const __m512i idx = _mm512_setr_epi64(0, 12, 24, 36, 48, 60, 72, 84);
    for (int i = 0; i < 12; i++) s[i] = canon8(_mm512_i64gather_epi64(idx, (const long long *)(states + i), 8));
    permute8(s);
    for (int i = 0; i < 12; i++) _mm512_i64scatter_epi64((long long *)(states + i), idx, s[i], 8);
}

/* leaves I .. I+7 of MerkleTreeWithCap::construct (merkle_tree.rs:78-174): cols[c][I..I+8) is one unaligned load */
AVX512 void orc_hash_leaves_x8(const uint64_t *const *cols, size_t n_cols, size_t I, uint64_t *out /* 8 digests, digest-major */) {
    v8 s[12];
    for (int i = 0; i < 12; i++) s[i] = _mm512_setzero_si512();
    size_t c = 0;
    while (n_cols - c >= 8) {                                  /* sponge.rs:224-346: overwrite the rate, permute */
        for (int k = 0; k < 8; k++) s[k] = canon8(_mm512_loadu_si512((const void *)(cols[c + k] + I)));
        permute8(s);
        c += 8;
    }
    if (c < n_cols) {
        const size_t rem = n_cols - c;
        for (size_t k = 0; k < rem; k++) s[k] = canon8(_mm512_loadu_si512((const void *)(cols[c + k] + I)));
        for (size_t k = rem; k < 8; k++) s[k] = _mm512_setzero_si512();
        permute8(s);
    }
    const __m512i idx = _mm512_setr_epi64(0, 4, 8, 12, 16, 20, 24, 28);
    for (int k = 0; k < 4; k++) _mm512_i64scatter_epi64((long long *)(out + k), idx, s[k], 8);
}
/* leaves I .. I+15 the same way, as two interleaved groups */
AVX512 void orc_hash_leaves_x16(const uint64_t *const *cols, size_t n_cols, size_t I, uint64_t *out /* 16 digests */) {
    v8 a[12], b[12];
    for (int i = 0; i < 12; i++) a[i] = b[i] = _mm512_setzero_si512();
    size_t c = 0;
    while (c < n_cols) {
        const size_t take = n_cols - c >= 8 ? 8 : n_cols - c;
        for (size_t k = 0; k < take; k++) {
            a[k] = canon8(_mm512_loadu_si512((const void *)(cols[c + k] + I)));
            b[k] = canon8(_mm512_loadu_si512((const void *)(cols[c + k] + I + 8)));
        }
        for (size_t k = take; k < 8; k++) a[k] = b[k] = _mm512_setzero_si512();
        permute8x2(a, b);
        c += take;
    }
    const __m512i idx = _mm512_setr_epi64(0, 4, 8, 12, 16, 20, 24, 28);
    for (int k = 0; k < 4; k++) {
        _mm512_i64scatter_epi64((long long *)(out + k), idx, a[k], 8);
        _mm512_i64scatter_epi64((long long *)(out + 32 + k), idx, b[k], 8);
    }
}
/* parents i .. i+7 of a node layer (oracle/mod.rs:162-168): children 2i, 2i+1 of `prev` (4 words each) */
AVX512 void orc_hash_nodes_x8(const uint64_t *prev, size_t i, uint64_t *next) {
    v8 s[12];
    const __m512i idx = _mm512_setr_epi64(0, 8, 16, 24, 32, 40, 48, 56);
    for (int k = 0; k < 8; k++) s[k] = canon8(_mm512_i64gather_epi64(idx, (const long long *)(prev + 8 * i + k), 8));
    for (int k = 8; k < 12; k++) s[k] = _mm512_setzero_si512();
    permute8(s);
    const __m512i odx = _mm512_setr_epi64(0, 4, 8, 12, 16, 20, 24, 28);
    for (int k = 0; k < 4; k++) _mm512_i64scatter_epi64((long long *)(next + 4 * i + k), odx, s[k], 8);
}
/* leaves j .. j+7 of construct_by_chunking (merkle_tree.rs:176-386): leaf j = hash(src0[jE..(j+1)E) || src1[jE..] || ...) */
AVX512 void orc_hash_chunked_x8(const uint64_t *const *srcs, size_t n_srcs, size_t E, size_t j, uint64_t *out) {
    v8 s[12];
    for (int i = 0; i < 12; i++) s[i] = _mm512_setzero_si512();
    const size_t total = n_srcs * E;
    const __m512i stride = _mm512_mullo_epi64(_mm512_setr_epi64(0, 1, 2, 3, 4, 5, 6, 7), _mm512_set1_epi64((long long)E));
    size_t e = 0;
    while (e < total) {
        size_t k = 0;
        for (; k < 8 && e < total; k++, e++)                   /* element e of the leaf: source e / E, offset e % E */
            s[k] = canon8(_mm512_i64gather_epi64(stride, (const long long *)(srcs[e / E] + j * E + e % E), 8));
        for (; k < 8; k++) s[k] = _mm512_setzero_si512();
        permute8(s);
    }
    const __m512i idx = _mm512_setr_epi64(0, 4, 8, 12, 16, 20, 24, 28);
    for (int k = 0; k < 4; k++) _mm512_i64scatter_epi64((long long *)(out + k), idx, s[k], 8);
}
