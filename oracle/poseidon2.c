/* ORACLE — TEST INFRASTRUCTURE ONLY (see gl.h).
 * Poseidon2 (t=12, rate 8, capacity 4, x^7, 4+22+4 rounds) over Goldilocks, the overwrite-mode sponge
 * used as tree hasher, and MerkleTreeWithCap, restated from the reference:
 *   permutation           implementations/poseidon2/state_generic_impl.rs:128-233
 *   external MDS          implementations/suggested_mds.rs:21-103  (circ(2*M4, M4, M4))
 *   internal diag shifts  implementations/poseidon2/params.rs:38-39
 *   sponge                algebraic_props/sponge.rs:224-346 (absorb = overwrite, zero-pad tail, no length tag)
 *   tree hasher           cs/oracle/mod.rs:114-176
 *   Merkle tree           cs/oracle/merkle_tree.rs:78-174 (construct), 176-386 (by chunking), 388-449 (nodes),
 *                         451-504 (cap / proof / verify)
 * Pinned bit-exactly by the reference's golden proof.json/vk.json (tests/test_oracle_fixture.py).
 * Trees are hashed eight leaves / eight parents at a time by poseidon2_avx512.c where the CPU has AVX-512 (the same algorithm
 * lane-wise; the reference's own fast path is AVX-512 too: implementations/poseidon2/state_avx512.rs); the code below is the
 * definition, the remainder loop, and the path of ORC_NO_AVX512=1.
 */
#include "oracle.h"
#include "poseidon_rc.h"
#include <stdlib.h>
#include <string.h>

static const uint64_t RC[BJ_POSEIDON_NUM_RC] = BJ_POSEIDON_RC_TABLE;
static const unsigned SH[12] = {4, 14, 11, 8, 0, 5, 2, 9, 13, 6, 3, 12};

static inline gl_t pow7(gl_t x) {               /* state_generic_impl.rs:142-149 */
    gl_t x2 = gl_sqr(x), x3 = gl_mul(x2, x), x4 = gl_sqr(x2);
    return gl_mul(x4, x3);
}
/* The linear layers are sums with small coefficients (external: entries of circ(2*M4, M4, M4) <= 14; internal: 1 + 2^k, k <= 14):
 * they are accumulated in 128-bit integers and reduced once per output word instead of once per addition. */
static inline void block_mul(const gl_t *x, u128 *y) {          /* suggested_mds.rs:21-56, on integers (y < 2^68) */
    u128 t0 = (u128)x[0] + x[1], t1 = (u128)x[2] + x[3];
    u128 t2 = 2 * (u128)x[1] + t1, t3 = 2 * (u128)x[3] + t0;
    u128 t4 = 4 * t1 + t3, t5 = 4 * t0 + t2;
    y[0] = t3 + t5; y[1] = t5; y[2] = t2 + t4; y[3] = t4;
}
static void ext_mds(gl_t *s) {                   /* suggested_mds.rs:59-103 */
    u128 y[12];
    block_mul(s, y); block_mul(s + 4, y + 4); block_mul(s + 8, y + 8);
    for (int j = 0; j < 4; j++) {
        u128 sum = y[j] + y[4 + j] + y[8 + j];                      /* < 2^70 */
        for (int b = 0; b < 3; b++) s[4 * b + j] = gl_reduce128(y[4 * b + j] + sum);
    }
}
static void full_round(gl_t *s, int r) {         /* state_generic_impl.rs:158-168 */
    for (int i = 0; i < 12; i++) s[i] = pow7(gl_add(s[i], gl_canon(RC[12 * r + i])));
    ext_mds(s);
}
static void partial_round(gl_t *s, int r) {      /* state_generic_impl.rs:171-219 */
    s[0] = pow7(gl_add(s[0], gl_canon(RC[12 * r])));
    u128 sum = 0;
    for (int i = 0; i < 12; i++) sum += s[i];                           /* < 2^68 */
    for (int i = 0; i < 12; i++) s[i] = gl_reduce128(((u128)s[i] << SH[i]) + sum);   /* < 2^79 */
}
void orc_poseidon2_permutation(uint64_t *s) {     /* state_generic_impl.rs:221-233 */
    for (int i = 0; i < 12; i++) s[i] = gl_canon(s[i]);
    ext_mds(s);
    int r = 0;
    for (int i = 0; i < 4; i++) full_round(s, r++);
    for (int i = 0; i < 22; i++) partial_round(s, r++);
    for (int i = 0; i < 4; i++) full_round(s, r++);
}

/* Poseidon (v1) permutation of the reference's naive implementation, used only by the GoldilocksPoisedonTranscript of
 * the SHA-256 bench script (implementations/poseidon_goldilocks_naive.rs:10-160): 4 full + 22 partial + 4 full rounds; every
 * round adds its 12 constants (ALL_ROUND_CONSTANTS, the same table Poseidon2 draws from), applies x^7 to all / to element 0,
 * and multiplies by the circulant MDS matrix M[row][col] = 2^EXPS[(col - row) mod 12].
 * PARITY UNPINNED: the reference holds no known-answer vector for this permutation (its tests compare the naive and the
 * optimised implementation, poseidon_goldilocks.rs:1036-1076), and its matrix is not Plonky2's; it is cross-checked
 * against an independent big-integer restatement (tests/test_poseidon1.py). */
static const unsigned MDS_EXPS[12] = {0, 0, 1, 0, 3, 5, 1, 8, 12, 3, 16, 10};
static void poseidon1_mds(gl_t *s) {
    gl_t out[12];
    for (int row = 0; row < 12; row++) {
        unsigned __int128 acc = 0;                                      /* < 2^81 (poseidon_goldilocks_naive.rs:37-60) */
        for (int col = 0; col < 12; col++) acc += (unsigned __int128)s[col] << MDS_EXPS[(col + 12 - row) % 12];
        uint64_t lo = (uint64_t)acc, hi = (uint64_t)(acc >> 64);      /* hi < 2^17: hi * 2^64 = hi * (2^32 - 1) */
        out[row] = gl_add(gl_canon(lo), gl_canon(hi * 0xFFFFFFFFULL));
    }
    memcpy(s, out, sizeof(out));
}
void orc_poseidon_permutation(uint64_t *s) {
    for (int i = 0; i < 12; i++) s[i] = gl_canon(s[i]);
    for (int r = 0; r < 30; r++) {
        for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], RC[12 * r + i]);
        const int full = r < 4 || r >= 26;
        for (int i = 0; i < (full ? 12 : 1); i++) s[i] = pow7(s[i]);
        poseidon1_mds(s);
    }
}

/* hash_into_leaf: sponge.rs:224-346 + oracle/mod.rs:141-151 */
void orc_hash_leaf(const uint64_t *els, size_t n, uint64_t *out4) {
    gl_t st[12] = {0};
    size_t i = 0;
    while (n - i >= 8) {
        for (int k = 0; k < 8; k++) st[k] = gl_canon(els[i + k]);
        orc_poseidon2_permutation(st);
        i += 8;
    }
    if (i < n) {
        size_t rem = n - i;
        for (size_t k = 0; k < rem; k++) st[k] = gl_canon(els[i + k]);
        for (size_t k = rem; k < 8; k++) st[k] = 0;
        orc_poseidon2_permutation(st);
    }
    memcpy(out4, st, 4 * sizeof(gl_t));
}
/* hash_into_node: oracle/mod.rs:162-168 */
void orc_hash_node(const uint64_t *l4, const uint64_t *r4, uint64_t *out4) {
    gl_t st[12] = {0};
    for (int k = 0; k < 4; k++) { st[k] = gl_canon(l4[k]); st[4 + k] = gl_canon(r4[k]); }
    orc_poseidon2_permutation(st);
    memcpy(out4, st, 4 * sizeof(gl_t));
}

/* continue_from_leaf_hashes (merkle_tree.rs:388-449).
 * `tree` holds all layers back to back: layer 0 = num_leaves digests, layer 1 = num_leaves/2, ... down to
 * the layer with cap_size digests (inclusive).  Total digests = 2*num_leaves - cap_size. */
size_t orc_merkle_tree_digests(size_t num_leaves, size_t cap_size) { return 2 * num_leaves - cap_size; }

void orc_merkle_nodes(uint64_t *tree, size_t num_leaves, size_t cap_size, int threads) {
    uint64_t *prev = tree;
    size_t len = num_leaves;
    while (len > cap_size) {
        uint64_t *next = prev + 4 * len;
        size_t nl = len / 2;
        const size_t nv = orc_poseidon2_avx512_available() ? nl / 8 * 8 : 0;     /* eight parents per AVX-512 call */
#pragma omp parallel for schedule(static) num_threads(threads)
        for (size_t i = 0; i < nv; i += 8) orc_hash_nodes_x8(prev, i, next);
#pragma omp parallel for schedule(static) num_threads(threads)
        for (size_t i = nv; i < nl; i++) orc_hash_node(prev + 8 * i, prev + 8 * i + 4, next + 4 * i);
        prev = next; len = nl;
    }
}

/* MerkleTreeWithCap::construct (merkle_tree.rs:78-174): cols[c] points at a column laid out [coset][n]
 * (= num_leaves contiguous values); leaf I = hash(cols[0][I], cols[1][I], ...). */
void orc_merkle_construct(const uint64_t *const *cols, size_t n_cols, size_t num_leaves, size_t cap_size,
                          uint64_t *tree, int threads) {
    const size_t nv = orc_poseidon2_avx512_available() ? num_leaves / 8 * 8 : 0;   /* eight leaves per AVX-512 call, lane = leaf */
    const size_t nv16 = nv / 16 * 16;                                              /* sixteen where they are there: two interleaved groups */
#pragma omp parallel for schedule(static) num_threads(threads)
    for (size_t I = 0; I < nv16; I += 16) orc_hash_leaves_x16(cols, n_cols, I, tree + 4 * I);
    if (nv16 < nv) orc_hash_leaves_x8(cols, n_cols, nv16, tree + 4 * nv16);
#pragma omp parallel num_threads(threads)
    {
        uint64_t *row = (uint64_t *)malloc(n_cols * sizeof(uint64_t));
#pragma omp for schedule(static)
        for (size_t I = nv; I < num_leaves; I++) {
            for (size_t c = 0; c < n_cols; c++) row[c] = cols[c][I];
            orc_hash_leaf(row, n_cols, tree + 4 * I);
        }
        free(row);
    }
    orc_merkle_nodes(tree, num_leaves, cap_size, threads);
}
/* same with the columns stored back to back: base[c*stride + I] */
void orc_merkle_construct_strided(const uint64_t *base, size_t stride, size_t n_cols, size_t num_leaves,
                                  size_t cap_size, uint64_t *tree, int threads) {
    const uint64_t **cols = (const uint64_t **)malloc(n_cols * sizeof(*cols));
    for (size_t c = 0; c < n_cols; c++) cols[c] = base + c * stride;
    orc_merkle_construct(cols, n_cols, num_leaves, cap_size, tree, threads);
    free(cols);
}

/* construct_by_chunking / _from_flat_sources (merkle_tree.rs:176-386): leaf j =
 * hash( src0[j*E .. (j+1)*E) || src1[j*E ..] || ... ),  E = elems_per_leaf; num_leaves = len / E */
void orc_merkle_construct_chunked(const uint64_t *const *srcs, size_t n_srcs, size_t len, size_t elems_per_leaf,
                                  size_t cap_size, uint64_t *tree, int threads) {
    size_t num_leaves = len / elems_per_leaf;
    const size_t nv = orc_poseidon2_avx512_available() ? num_leaves / 8 * 8 : 0;
#pragma omp parallel for schedule(static) num_threads(threads)
    for (size_t j = 0; j < nv; j += 8) orc_hash_chunked_x8(srcs, n_srcs, elems_per_leaf, j, tree + 4 * j);
#pragma omp parallel num_threads(threads)
    {
        uint64_t *row = (uint64_t *)malloc(n_srcs * elems_per_leaf * sizeof(uint64_t));
#pragma omp for schedule(static)
        for (size_t j = nv; j < num_leaves; j++) {
            for (size_t s = 0; s < n_srcs; s++)
                memcpy(row + s * elems_per_leaf, srcs[s] + j * elems_per_leaf, elems_per_leaf * sizeof(uint64_t));
            orc_hash_leaf(row, n_srcs * elems_per_leaf, tree + 4 * j);
        }
        free(row);
    }
    orc_merkle_nodes(tree, num_leaves, cap_size, threads);
}

/* get_cap: merkle_tree.rs:451-460 */
void orc_merkle_cap(const uint64_t *tree, size_t num_leaves, size_t cap_size, uint64_t *cap_out) {
    memcpy(cap_out, tree + 4 * (2 * num_leaves - 2 * cap_size), 4 * cap_size * sizeof(uint64_t));
}
/* get_proof: merkle_tree.rs:462-480. path_out receives depth*4 words, depth = log2(num_leaves/cap_size) */
size_t orc_merkle_proof(const uint64_t *tree, size_t num_leaves, size_t cap_size, size_t idx,
                        uint64_t *leaf_hash_out, uint64_t *path_out) {
    const uint64_t *layer = tree;
    size_t len = num_leaves, depth = 0;
    memcpy(leaf_hash_out, tree + 4 * idx, 32);
    while (len > cap_size) {
        memcpy(path_out + 4 * depth, layer + 4 * (idx ^ 1), 32);
        layer += 4 * len; len /= 2; idx >>= 1; depth++;
    }
    return depth;
}
/* verify_proof_over_cap: merkle_tree.rs:482-504 */
int orc_merkle_verify(const uint64_t *path, size_t depth, const uint64_t *cap, const uint64_t *leaf_hash, size_t idx) {
    uint64_t cur[4], nxt[4];
    for (int k = 0; k < 4; k++) cur[k] = gl_canon(leaf_hash[k]);
    for (size_t d = 0; d < depth; d++) {
        if ((idx & 1) == 0) orc_hash_node(cur, path + 4 * d, nxt);
        else orc_hash_node(path + 4 * d, cur, nxt);
        memcpy(cur, nxt, 32);
        idx >>= 1;
    }
    for (int k = 0; k < 4; k++) if (gl_canon(cap[4 * idx + k]) != cur[k]) return 0;
    return 1;
}
