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
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gl {

typedef uint64_t u64;
typedef uint32_t u32;

static constexpr u64 P = 0xFFFFFFFF00000001ULL;
static constexpr u64 EPS = 0xFFFFFFFFULL;  // 2^64 mod p
static constexpr u64 GEN = 7;              // multiplicative generator = LDE coset shift = F_p^2 non-residue

__host__ __device__ __forceinline__ u64 canon(u64 a) { return a >= P ? a - P : a; }

// a, b in [0,p) -> (a + b) mod p
__host__ __device__ __forceinline__ u64 add(u64 a, u64 b) {
    u64 s = a + b;
    u64 t = s + EPS;  // s - p (mod 2^64)
    // s >= p  <=>  s + EPS wraps; a + b wrapped <=> s < a
    return (s < a || t < s) ? t : s;
}
// a, b in [0,p) -> (a - b) mod p
__host__ __device__ __forceinline__ u64 sub(u64 a, u64 b) {
    u64 d = a - b;
    return (a < b) ? d - EPS : d;  // + p == - EPS (mod 2^64)
}
__host__ __device__ __forceinline__ u64 neg(u64 a) { return a ? P - a : 0; }
__host__ __device__ __forceinline__ u64 dbl(u64 a) { return add(a, a); }

// (hi:lo) 128-bit -> canonical residue
__host__ __device__ __forceinline__ u64 reduce128(u64 hi, u64 lo) {
    u64 hi_hi = hi >> 32, hi_lo = hi & EPS;
    u64 t0 = lo - hi_hi;
    if (lo < hi_hi) t0 -= EPS;
    u64 t1 = (hi_lo << 32) - hi_lo;  // hi_lo * (2^32 - 1), no multiplier needed
    u64 r = t0 + t1;
    if (r < t1) r += EPS;
    return canon(r);
}

__host__ __device__ __forceinline__ void mul_wide(u64 a, u64 b, u64 &hi, u64 &lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0;
    u64 mid = (u64)a0 * b1 + (p00 >> 32);        // v_mad_u64_u32, cannot overflow
    u64 mid2 = (u64)a1 * b0 + (mid & EPS);       // v_mad_u64_u32, cannot overflow
    hi = (u64)a1 * b1 + (mid >> 32) + (mid2 >> 32);
    lo = (mid2 << 32) | (p00 & EPS);
#else
    unsigned __int128 x = (unsigned __int128)a * b;
    hi = (u64)(x >> 64);
    lo = (u64)x;
#endif
}

__host__ __device__ __forceinline__ u64 mul(u64 a, u64 b) {
    u64 hi, lo;
    mul_wide(a, b, hi, lo);
    return reduce128(hi, lo);
}
__host__ __device__ __forceinline__ u64 sqr(u64 a) { return mul(a, a); }

// a * 2^k mod p for 0 <= k < 64 (shift instead of multiply; used by Poseidon2's internal matrix)
__host__ __device__ __forceinline__ u64 mul_pow2(u64 a, unsigned k) {
    if (k == 0) return a;
    return reduce128(a >> (64 - k), a << k);
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
