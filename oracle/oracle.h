/* ORACLE — TEST INFRASTRUCTURE ONLY.
 * CPU restatement of the reference's (matter-labs/era-boojum, pure Rust, cannot be built here — no Rust
 * toolchain) algorithms for the proving hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / reported baseline.  The product
 * (era_boojum_amd/, include/boojum_hip.h) never links or calls it.
 *
 * Parity status: PINNED for Poseidon2 / sponge / Merkle / Poseidon2 transcript / query-index extraction /
 * FRI fold / DEEP quotient by the reference's own golden proof.json + vk.json (tests/golden/, checked in
 * tests/test_oracle_fixture.py).  NTT/LDE conventions are pinned indirectly (FRI final-monomial check of the
 * fixture) plus the reference's own differential properties (NTT == naive DFT, iNTT∘NTT = id).
 */
#ifndef ORACLE_H
#define ORACLE_H
#include "gl.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- ntt.c ---- */
void orc_twiddles(uint64_t *out, unsigned log_n, int inverse);
void orc_bitreverse(uint64_t *a, unsigned log_n);
void orc_canonicalize(uint64_t *a, size_t n);
void orc_fft_natural_to_bitreversed(uint64_t *a, unsigned log_n, uint64_t coset, const uint64_t *tw);
void orc_ifft_natural_to_natural(uint64_t *a, unsigned log_n, uint64_t coset, const uint64_t *inv_tw);
void orc_naive_dft(const uint64_t *a, uint64_t *out, unsigned log_n, uint64_t coset);
void orc_lde_coset_shifts(uint64_t *out, unsigned log_n, unsigned log_lde);
void orc_lde_from_monomials(const uint64_t *mono, uint64_t *out, unsigned log_n, unsigned log_lde, const uint64_t *fwd_tw);
void orc_fft_batch(uint64_t *cols, unsigned log_n, size_t n_cols, uint64_t coset, int threads);
void orc_fft_batch_to(const uint64_t *src, uint64_t *dst, unsigned log_n, size_t n_cols, uint64_t coset, int threads);
void orc_ifft_batch(uint64_t *cols, unsigned log_n, size_t n_cols, uint64_t coset, int threads);
void orc_lde_batch(const uint64_t *mono, uint64_t *out, unsigned log_n, unsigned log_lde, size_t n_cols, int threads);

/* ---- poseidon2.c ---- */
void orc_poseidon2_permutation(uint64_t *state12);
void orc_poseidon_permutation(uint64_t *state12);   /* Poseidon v1 (naive), bench-script transcript only; parity unpinned */
void orc_hash_leaf(const uint64_t *els, size_t n, uint64_t *out4);
void orc_hash_node(const uint64_t *l4, const uint64_t *r4, uint64_t *out4);
size_t orc_merkle_tree_digests(size_t num_leaves, size_t cap_size);
void orc_merkle_nodes(uint64_t *tree, size_t num_leaves, size_t cap_size, int threads);
void orc_merkle_construct(const uint64_t *const *cols, size_t n_cols, size_t num_leaves, size_t cap_size, uint64_t *tree, int threads);
void orc_merkle_construct_strided(const uint64_t *base, size_t stride, size_t n_cols, size_t num_leaves, size_t cap_size, uint64_t *tree, int threads);
void orc_merkle_construct_chunked(const uint64_t *const *srcs, size_t n_srcs, size_t len, size_t elems_per_leaf, size_t cap_size, uint64_t *tree, int threads);
void orc_merkle_cap(const uint64_t *tree, size_t num_leaves, size_t cap_size, uint64_t *cap_out);
size_t orc_merkle_proof(const uint64_t *tree, size_t num_leaves, size_t cap_size, size_t idx, uint64_t *leaf_hash_out, uint64_t *path_out);
int orc_merkle_verify(const uint64_t *path, size_t depth, const uint64_t *cap, const uint64_t *leaf_hash, size_t idx);

/* ---- poseidon2_avx512.c (eight permutations lane-wise; run-time selected) ---- */
int orc_poseidon2_avx512_available(void);           /* 1: the CPU has AVX-512 F + DQ and ORC_NO_AVX512 is not set */
void orc_poseidon2_permutation_x8(uint64_t *states96);   /* eight states back to back; call only when available */
void orc_hash_leaves_x8(const uint64_t *const *cols, size_t n_cols, size_t I, uint64_t *out32);
void orc_hash_leaves_x16(const uint64_t *const *cols, size_t n_cols, size_t I, uint64_t *out64);   /* two interleaved groups of eight */
void orc_hash_nodes_x8(const uint64_t *prev, size_t i, uint64_t *next);
void orc_hash_chunked_x8(const uint64_t *const *srcs, size_t n_srcs, size_t E, size_t j, uint64_t *out32);

/* ---- transcript.c ---- */
typedef struct orc_transcript orc_transcript;
orc_transcript *orc_transcript_new(void);
orc_transcript *orc_transcript_new_kind(int kind);   /* 1 = Poseidon2, 2 = Poseidon (v1) */
void orc_transcript_free(orc_transcript *t);
void orc_transcript_absorb(orc_transcript *t, const uint64_t *els, size_t n);
uint64_t orc_transcript_challenge(orc_transcript *t);
/* BoolsBuffer-based query index extraction (transcript.rs:369-417, prover.rs:2161-2182) */
typedef struct orc_bools orc_bools;
orc_bools *orc_bools_new(unsigned max_needed_bits);
void orc_bools_free(orc_bools *b);
uint64_t orc_query_index(orc_bools *b, orc_transcript *t, unsigned log_n, unsigned log_lde);

/* ---- fri.c ---- */
/* returns schedule length; out_sched must hold >= 32 entries */
size_t orc_fri_schedule(uint32_t security_bits, size_t cap_size, uint32_t pow_bits, uint32_t rate_log2,
                        uint32_t initial_degree_log2, uint32_t *new_pow_bits, size_t *num_queries,
                        uint32_t *out_sched, size_t *final_degree);
void orc_fri_fold(const uint64_t *c0, const uint64_t *c1, size_t len, uint64_t *o0, uint64_t *o1,
                  const uint64_t *roots, uint64_t coset_inv, uint64_t ch0, uint64_t ch1);
typedef struct {
    size_t num_oracles;             /* base + intermediates = schedule length */
    uint64_t *trees[32];            /* all layers of each oracle (see orc_merkle_nodes layout) */
    size_t tree_leaves[32];
    size_t elems_per_leaf[32];      /* 2^schedule[i] values of c0 then of c1 */
    uint64_t *src_c0[32], *src_c1[32]; /* leaf sources of oracle i (i=0: the input codeword, not owned) */
    size_t src_len[32];
    uint64_t *final_c0, *final_c1;  /* final monomials */
    size_t final_degree;
    uint64_t challenges[32][2];     /* first folding challenge drawn for each step */
} orc_fri_result;
orc_fri_result *orc_do_fri(const uint64_t *c0, const uint64_t *c1, unsigned log_full, unsigned log_lde,
                           const uint32_t *schedule, size_t sched_len, size_t cap_size,
                           orc_transcript *t, int threads);
void orc_fri_result_free(orc_fri_result *r);

/* ---- pointwise.c ---- */
void orc_batch_inverse(const uint64_t *in, uint64_t *out, size_t n);
void orc_ext_batch_inverse(const uint64_t *in0, const uint64_t *in1, uint64_t *o0, uint64_t *o1, size_t n);
void orc_barycentric_weights(unsigned log_n, uint64_t coset, const uint64_t *at, uint64_t *w0, uint64_t *w1);
void orc_barycentric_eval_base(const uint64_t *values, const uint64_t *w0, const uint64_t *w1, size_t n, uint64_t *out2);
void orc_barycentric_eval_ext(const uint64_t *v0, const uint64_t *v1, const uint64_t *w0, const uint64_t *w1, size_t n, uint64_t *out2);
void orc_deep_quotient_point(const uint64_t *f0, const uint64_t *f1, const unsigned char *is_ext, size_t n_src,
                             const uint64_t *values, const uint64_t *challenges, const uint64_t *at, uint64_t x,
                             uint64_t *out2);
void orc_deep_quotient_accumulate(const uint64_t *const *src_c0, const uint64_t *const *src_c1, size_t n_src,
                                  const uint64_t *values, const uint64_t *challenges, const uint64_t *at, unsigned log_n,
                                  unsigned log_lde, uint64_t *dst0, uint64_t *dst1, int threads);
void orc_deep_quotient_accumulate_range(const uint64_t *const *src_c0, const uint64_t *const *src_c1, size_t n_src,
                                        const uint64_t *values, const uint64_t *challenges, const uint64_t *at, unsigned log_n,
                                        unsigned log_lde, size_t first, size_t count, uint64_t *dst0, uint64_t *dst1, int threads);

#ifdef __cplusplus
}
#endif
#endif
