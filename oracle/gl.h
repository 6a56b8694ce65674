/* ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ may be imported, linked or executed by the
 * product path (era_boojum_amd/, include/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker.
 *
 * Goldilocks field p = 2^64 - 2^32 + 1 and its quadratic extension F_p[u]/(u^2 - 7), restated in
 * portable C from the reference:
 *   - field constants / reduction:  src/field/goldilocks/mod.rs:109-117, 188-201 (from_u128_with_reduction),
 *     215-255 (add/mul), 309 (sub)
 *   - extension:  src/field/goldilocks/extension.rs:14-16 (NON_RESIDUE = 7),
 *     src/field/traits/field.rs:407-426 (mul), 484-512 (inverse through the norm)
 * The reference allows non-canonical u64 in memory and reduces on compare/serialise; the oracle keeps
 * every value canonical (in [0,p)) at all times, so parity is "equality of canonical residues".
 */
#ifndef ORACLE_GL_H
#define ORACLE_GL_H
#include <stdint.h>
#include <stddef.h>

typedef uint64_t gl_t;
typedef unsigned __int128 u128;
#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL /* 2^64 mod p */
#define GL_GEN 7ULL           /* multiplicative generator, LDE coset shift, and the F_p^2 non-residue */
#define GL_OMEGA_2_32 0x185629dcda58878cULL /* radix_2_subgroup_generator, goldilocks/mod.rs:109-117 */

/* written without data-dependent branches (the conditions are coin flips on random field elements: compilers turn these
 * forms into conditional moves / masks, mispredicted branches cost more than the arithmetic) */
static inline gl_t gl_canon(gl_t a) {
    gl_t t = a - GL_P;
    return a >= GL_P ? t : a;
}

static inline gl_t gl_add(gl_t a, gl_t b) { /* canonical in, canonical out */
    gl_t s = a + b;
    gl_t t = s - GL_P;                      /* = s + EPS mod 2^64: the value when the sum wrapped or reached p */
    return ((s < a) | (s >= GL_P)) ? t : s;
}
static inline gl_t gl_sub(gl_t a, gl_t b) {
    gl_t d = a - b;
    return d - ((0 - (gl_t)(a < b)) & GL_EPS);   /* borrowed 2^64 = p + EPS: give EPS back */
}
static inline gl_t gl_neg(gl_t a) { return a ? GL_P - a : 0; }
static inline gl_t gl_dbl(gl_t a) { return gl_add(a, a); }

/* 128-bit -> canonical residue: 2^64 = 2^32 - 1, 2^96 = -1 (mod p)  (mod.rs:188-201) */
static inline gl_t gl_reduce128(u128 x) {
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    uint64_t hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
    uint64_t t0 = lo - hi_hi;
    t0 -= (0 - (uint64_t)(lo < hi_hi)) & GL_EPS;   /* borrow: subtract 2^64 mod p */
    uint64_t t1 = hi_lo * GL_EPS;                  /* < 2^64 */
    uint64_t r = t0 + t1;
    r += (0 - (uint64_t)(r < t1)) & GL_EPS;        /* carry */
    return gl_canon(r);
}
static inline gl_t gl_mul(gl_t a, gl_t b) { return gl_reduce128((u128)a * b); }
static inline gl_t gl_sqr(gl_t a) { return gl_mul(a, a); }
static inline gl_t gl_pow(gl_t a, uint64_t e) {
    gl_t r = 1;
    while (e) { if (e & 1) r = gl_mul(r, a); a = gl_sqr(a); e >>= 1; }
    return r;
}
static inline gl_t gl_inv(gl_t a) { return gl_pow(a, GL_P - 2); }
static inline gl_t gl_from_u64(uint64_t a) { return a % GL_P; }

/* ---- F_p^2 = F_p[u]/(u^2-7) ---- */
typedef struct { gl_t c0, c1; } gl2_t;
static inline gl2_t gl2_make(gl_t c0, gl_t c1) { gl2_t r = {c0, c1}; return r; }
static inline gl2_t gl2_add(gl2_t a, gl2_t b) { return gl2_make(gl_add(a.c0, b.c0), gl_add(a.c1, b.c1)); }
static inline gl2_t gl2_sub(gl2_t a, gl2_t b) { return gl2_make(gl_sub(a.c0, b.c0), gl_sub(a.c1, b.c1)); }
static inline gl2_t gl2_neg(gl2_t a) { return gl2_make(gl_neg(a.c0), gl_neg(a.c1)); }
static inline gl2_t gl2_mul(gl2_t a, gl2_t b) {
    gl_t v0 = gl_mul(a.c0, b.c0), v1 = gl_mul(a.c1, b.c1);
    gl_t c1 = gl_sub(gl_sub(gl_mul(gl_add(a.c0, a.c1), gl_add(b.c0, b.c1)), v0), v1);
    return gl2_make(gl_add(v0, gl_mul(v1, GL_GEN)), c1);
}
static inline gl2_t gl2_sqr(gl2_t a) { return gl2_mul(a, a); }
static inline gl2_t gl2_mul_base(gl2_t a, gl_t s) { return gl2_make(gl_mul(a.c0, s), gl_mul(a.c1, s)); }
static inline gl2_t gl2_inv(gl2_t a) {
    gl_t norm = gl_sub(gl_sqr(a.c0), gl_mul(GL_GEN, gl_sqr(a.c1)));
    gl_t ni = gl_inv(norm);
    return gl2_make(gl_mul(a.c0, ni), gl_neg(gl_mul(a.c1, ni)));
}
static inline gl2_t gl2_pow(gl2_t a, uint64_t e) {
    gl2_t r = gl2_make(1, 0);
    while (e) { if (e & 1) r = gl2_mul(r, a); a = gl2_sqr(a); e >>= 1; }
    return r;
}

static inline uint64_t bitrev64(uint64_t x, unsigned bits) {
    uint64_t r = 0;
    for (unsigned i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}
/* domain_generator_for_size: cs/implementations/utils.rs:13-28 */
static inline gl_t gl_omega(unsigned log_n) {
    gl_t w = GL_OMEGA_2_32;
    for (unsigned i = log_n; i < 32; i++) w = gl_sqr(w);
    return w;
}
#endif
