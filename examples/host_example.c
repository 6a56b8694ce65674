/* A foreign-language host in miniature: plain C99 against include/boojum_hip.h and libboojum_hip.so, nothing else
 * (what a cgo / Rust-FFI / JNI binding does).  It commits to a batch of columns the way one prover round
 * does — inverse NTT to monomials, low-degree extension, Poseidon2 Merkle tree, cap — and checks what it can check
 * without a second implementation:
 *   - iNTT followed by the forward NTT on the base coset is the identity up to the bit-reversed output order,
 *   - a Merkle path returned by the library hashes up to the cap entry it belongs to (node hashing redone with
 *     bj_poseidon2_permute),
 *   - the transcript hands out the same challenges for the same absorbed cap.
 * Exit code 0 = all checks passed, 2 = no GPU (the library has no CPU fallback), 1 = a check failed.
 *   gcc -std=c99 -Iinclude examples/host_example.c -o host_example -Lera_boojum_amd -lboojum_hip -Wl,-rpath,$PWD/era_boojum_amd
 */
#include "boojum_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define P 0xFFFFFFFF00000001ULL
#define CHECK(call)                                                                                  \
    do {                                                                                             \
        int rc_ = (call);                                                                            \
        if (rc_ != 0) {                                                                              \
            fprintf(stderr, "%s -> %s (%s)\n", #call, bj_status_string(rc_), bj_last_error(ctx));   \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)

static uint64_t splitmix(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static unsigned bitrev(unsigned x, unsigned bits) {
    unsigned r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

int main(void) {
    enum { LOG_N = 12, COLS = 11, LOG_LDE = 3, CAP = 16 };
    const size_t n = (size_t)1 << LOG_N, L = (size_t)1 << LOG_LDE, leaves = n * L;
    bj_ctx *ctx = NULL;
    if (bj_device_count() <= 0 || bj_ctx_create(0, &ctx) != 0) {
        fprintf(stderr, "no HIP device: this library has no CPU path\n");
        return 2;
    }
    printf("ABI version %d\n", bj_abi_version());

    /* trace columns, natural order, canonical residues */
    uint64_t *h_cols = (uint64_t *)malloc(COLS * n * 8), *h_back = (uint64_t *)malloc(COLS * n * 8);
    uint64_t seed = 20240807;
    for (size_t i = 0; i < COLS * n; i++) h_cols[i] = splitmix(&seed) % P;
    void *d_cols = NULL, *d_mono = NULL, *d_lde = NULL, *d_tree = NULL, *d_states = NULL;
    CHECK(bj_malloc(ctx, COLS * n * 8, &d_cols));
    CHECK(bj_malloc(ctx, COLS * n * 8, &d_mono));
    CHECK(bj_malloc(ctx, COLS * leaves * 8, &d_lde));
    CHECK(bj_malloc(ctx, bj_merkle_tree_digests(leaves, CAP) * 32, &d_tree));
    CHECK(bj_memcpy_h2d(ctx, d_cols, h_cols, COLS * n * 8));

    /* 1. evaluations -> monomials -> evaluations */
    CHECK(bj_intt_batch(ctx, (const uint64_t *)d_cols, (uint64_t *)d_mono, LOG_N, COLS, n, 1));
    CHECK(bj_ntt_forward_batch(ctx, (const uint64_t *)d_mono, (uint64_t *)d_cols, LOG_N, COLS, n, 1));
    CHECK(bj_memcpy_d2h(ctx, h_back, d_cols, COLS * n * 8));
    for (size_t c = 0; c < COLS; c++)
        for (size_t i = 0; i < n; i++)
            if (h_back[c * n + bitrev((unsigned)i, LOG_N)] != h_cols[c * n + i]) {
                fprintf(stderr, "NTT round trip differs at column %zu, row %zu\n", c, i);
                return 1;
            }
    printf("iNTT -> NTT round trip: %d columns x 2^%d ok\n", COLS, LOG_N);

    /* 2. commit: LDE + tree + cap, then open one leaf and walk its path */
    CHECK(bj_lde_batch(ctx, (const uint64_t *)d_mono, n, (uint64_t *)d_lde, LOG_N, COLS, LOG_LDE));
    CHECK(bj_merkle_tree_build(ctx, (const uint64_t *)d_lde, leaves, COLS, leaves, CAP, (uint64_t *)d_tree));
    uint64_t cap[CAP * 4], leaf[4], path[32 * 4];
    CHECK(bj_merkle_tree_cap(ctx, (const uint64_t *)d_tree, leaves, CAP, cap));
    const size_t idx = 12345 % leaves;
    unsigned depth = 0;
    while (((size_t)CAP << depth) < leaves) depth++;
    CHECK(bj_merkle_tree_proof(ctx, (const uint64_t *)d_tree, leaves, CAP, idx, leaf, path));
    CHECK(bj_malloc(ctx, 12 * 8, &d_states));
    uint64_t cur[4], state[12];
    memcpy(cur, leaf, 32);
    size_t pos = idx;
    for (unsigned d = 0; d < depth; d++, pos >>= 1) {   /* node = permutation(left || right || 0000)[0..4] */
        const uint64_t *sib = path + 4 * d;
        memcpy(state, (pos & 1) ? sib : cur, 32);
        memcpy(state + 4, (pos & 1) ? cur : sib, 32);
        memset(state + 8, 0, 32);
        CHECK(bj_memcpy_h2d(ctx, d_states, state, sizeof state));
        CHECK(bj_poseidon2_permute(ctx, (uint64_t *)d_states, 1));
        CHECK(bj_memcpy_d2h(ctx, state, d_states, sizeof state));
        memcpy(cur, state, 32);
    }
    if (memcmp(cur, cap + 4 * pos, 32) != 0) {
        fprintf(stderr, "Merkle path of leaf %zu does not reach cap entry %zu\n", idx, pos);
        return 1;
    }
    printf("Merkle tree over %zu leaves x %d columns: path of leaf %zu reaches cap[%zu]\n", leaves, COLS, idx, pos);

    /* 3. Fiat-Shamir: same cap, same challenges */
    uint64_t ch[2][2];
    for (int k = 0; k < 2; k++) {
        bj_transcript *t = NULL;
        if (bj_transcript_create(BJ_TRANSCRIPT_POSEIDON2, &t) != 0) return 1;
        if (bj_transcript_absorb_cap(t, cap, CAP * 4) || bj_transcript_challenge(t, &ch[k][0]) ||
            bj_transcript_challenge(t, &ch[k][1]))
            return 1;
        bj_transcript_destroy(t);
    }
    if (ch[0][0] != ch[1][0] || ch[0][1] != ch[1][1] || ch[0][0] >= P) {
        fprintf(stderr, "transcript is not deterministic\n");
        return 1;
    }
    printf("transcript challenge after the cap: %016llx %016llx\n", (unsigned long long)ch[0][0], (unsigned long long)ch[0][1]);

    bj_free(ctx, d_states); bj_free(ctx, d_tree); bj_free(ctx, d_lde); bj_free(ctx, d_mono); bj_free(ctx, d_cols);
    bj_ctx_destroy(ctx);
    free(h_cols); free(h_back);
    printf("ok\n");
    return 0;
}
