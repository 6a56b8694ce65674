/* Host-side Goldilocks vector arithmetic for INPUT SYNTHESIS only (era_boojum_amd/synthetic.py builds the satisfiable
 * SHA-shaped circuits the benches and tests prove).  Not on the proving path and not part of libboojum_hip.so: the
 * numpy implementation in field_np.py computes the same thing, this one is ~50x faster so that a 2^22-row circuit is
 * generated in seconds.  Built by era_boojum_amd/build.py with gcc -O3 -fopenmp. */
#include <omp.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

/* cap the OpenMP team: the default (one thread per host core, 256 on the GPU boxes) costs more in fork/join than the
 * loops below take */
void synth_set_threads(int n) { omp_set_num_threads(n < 1 ? 1 : n); }

#define GL_P 0xFFFFFFFF00000001ULL

static inline uint64_t gl_canon(uint64_t a) { return a >= GL_P ? a - GL_P : a; }

static inline uint64_t gl_mul(uint64_t a, uint64_t b) {
    unsigned __int128 w = (unsigned __int128)a * b;
    uint64_t lo = (uint64_t)w, hi = (uint64_t)(w >> 64);
    uint64_t hh = hi >> 32, hl = hi & 0xFFFFFFFFULL;
    /* 2^64 = 2^32 - 1, 2^96 = -1 (mod p) */
    uint64_t t = lo - hh;
    if (lo < hh) t -= 0xFFFFFFFFULL;
    uint64_t m = hl * 0xFFFFFFFFULL;
    uint64_t r = t + m;
    if (r < m) r += 0xFFFFFFFFULL;
    return gl_canon(r);
}

static inline uint64_t gl_add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    if (s < a) s += 0xFFFFFFFFULL;
    return gl_canon(s);
}

void synth_mul(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n) {
#pragma omp parallel for schedule(static) if (n > 65536)
    for (size_t i = 0; i < n; i++) out[i] = gl_mul(gl_canon(a[i]), gl_canon(b[i]));
}

void synth_mul_scalar(const uint64_t *a, uint64_t s, uint64_t *out, size_t n) {
    s = gl_canon(s);
#pragma omp parallel for schedule(static) if (n > 65536)
    for (size_t i = 0; i < n; i++) out[i] = gl_mul(gl_canon(a[i]), s);
}

/* out = a*b + c*d */
void synth_fma2(const uint64_t *a, const uint64_t *b, const uint64_t *c, const uint64_t *d, uint64_t *out, size_t n) {
#pragma omp parallel for schedule(static) if (n > 65536)
    for (size_t i = 0; i < n; i++)
        out[i] = gl_add(gl_mul(gl_canon(a[i]), gl_canon(b[i])), gl_mul(gl_canon(c[i]), gl_canon(d[i])));
}

/* out[i] = base^(start + i) */
void synth_powers(uint64_t base, uint64_t *out, size_t n) {
    base = gl_canon(base);
    if (!n) return;
    const size_t B = 4096;
#pragma omp parallel for schedule(static)
    for (size_t blk = 0; blk < (n + B - 1) / B; blk++) {
        size_t i0 = blk * B, e = i0;
        uint64_t cur = 1, sq = base;
        while (e) {
            if (e & 1) cur = gl_mul(cur, sq);
            sq = gl_mul(sq, sq);
            e >>= 1;
        }
        for (size_t i = i0; i < i0 + B && i < n; i++) {
            out[i] = cur;
            cur = gl_mul(cur, base);
        }
    }
}

/* ---- helpers of era_boojum_amd/sha256_circuit.py (the real SHA-256 circuit of the reference's bench) ---- */

/* Copy-permutation polynomials from the variable placement (what create_permutation_polys computes,
 * src/cs/implementations/setup.rs:419-503): walking the cells column by column, row by row, every cell of a variable
 * receives the identity k_col * omega^row of the variable's previous cell, and its first cell that of its last one.
 * var_ids[col][row] < 0 marks an empty cell (sigma = identity there).  sigma must come in holding the identities. */
void synth_sigma_from_placement(const int32_t *var_ids, size_t num_cols, size_t n, size_t num_vars, uint64_t *sigma) {
    uint64_t *prev = (uint64_t *)malloc(num_vars * sizeof(uint64_t));
    uint64_t *first = (uint64_t *)malloc(num_vars * sizeof(uint64_t));
    const uint64_t NONE = ~0ULL;
    for (size_t v = 0; v < num_vars; v++) first[v] = NONE;
    for (size_t cell = 0; cell < num_cols * n; cell++) {
        int32_t v = var_ids[cell];
        if (v < 0) continue;
        if (first[v] == NONE) {
            first[v] = cell;
            prev[v] = sigma[cell];
        } else {
            uint64_t here = sigma[cell];
            sigma[cell] = prev[v];
            prev[v] = here;
        }
    }
    for (size_t v = 0; v < num_vars; v++)
        if (first[v] != NONE) sigma[first[v]] = prev[v];
    free(prev);
    free(first);
}

/* chaining values of SHA-256 (FIPS 180-4) before every block: states[(b+1)*8 ..] = compress(states[b*8 ..], block b) */
static inline uint32_t rotr32(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }
void synth_sha256_states(const uint8_t *padded, size_t num_blocks, const uint32_t *round_constants, uint32_t *states) {
    for (size_t b = 0; b < num_blocks; b++) {
        uint32_t w[64];
        const uint8_t *m = padded + 64 * b;
        for (int i = 0; i < 16; i++)
            w[i] = ((uint32_t)m[4 * i] << 24) | ((uint32_t)m[4 * i + 1] << 16) | ((uint32_t)m[4 * i + 2] << 8) | m[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            uint32_t s0 = rotr32(w[i - 15], 7) ^ rotr32(w[i - 15], 18) ^ (w[i - 15] >> 3);
            uint32_t s1 = rotr32(w[i - 2], 17) ^ rotr32(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        const uint32_t *h = states + 8 * b;
        uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            uint32_t t1 = hh + (rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25)) + ((e & f) ^ (~e & g)) + round_constants[i] + w[i];
            uint32_t t2 = (rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22)) + ((a & bb) ^ (a & c) ^ (bb & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
        }
        uint32_t *o = states + 8 * (b + 1);
        o[0] = h[0] + a; o[1] = h[1] + bb; o[2] = h[2] + c; o[3] = h[3] + d;
        o[4] = h[4] + e; o[5] = h[5] + f; o[6] = h[6] + g; o[7] = h[7] + hh;
    }
}

/* variables[cell] = value of the variable placed in the cell, 0 for an empty cell (witness.rs:325-385) */
void synth_gather_values(const int32_t *var_ids, const uint64_t *vals, uint64_t *out, size_t cells) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < cells; i++) out[i] = var_ids[i] >= 0 ? vals[var_ids[i]] : 0;
}
