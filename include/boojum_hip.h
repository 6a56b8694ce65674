/* boojum_hip.h — C ABI of the MI355X-native Boojum proving hot path (libboojum_hip.so).
 *
 * The reference (matter-labs/era-boojum, pure Rust) has no FFI boundary of its own; this
 * header IS the boundary a Rust host binds with `extern "C"` (see INTEGRATION.md for the shim).  Every entry point
 * names the reference interface it replaces (paths relative to the reference root).
 *
 * Conventions
 *  - All field data are Goldilocks elements (p = 2^64 - 2^32 + 1) stored as little-endian u64.  Inputs may be any
 *    u64 (the reference tolerates non-canonical values in memory, src/field/goldilocks/mod.rs:98-107); every output
 *    is the canonical residue in [0, p).  Parity with the reference = equality of canonical residues.
 *  - F_p^2 = F_p[u]/(u^2-7) values are two separate base columns (c0, c1), never interleaved
 *    (src/field/traits/field_like.rs:617-642).
 *  - "d_" pointers are device (HBM) pointers on the context's GPU, "h_" pointers are host pointers.  Plain
 *    pointers and sizes only; no torch / C++ types cross this boundary.
 *  - Column batches are column-major: column c starts at base + c*col_stride (in elements).  An LDE column is
 *    [coset][n] with each coset in bit-reversed order, i.e. global index I = coset*n + i is the bit-reversed
 *    enumeration of g*<w_{nL}> (src/cs/implementations/utils.rs:311-403, src/cs/implementations/proof.rs:89-91).
 *  - Work is enqueued on the context's HIP stream and is asynchronous unless stated; bj_sync() drains it.
 *  - Errors: functions return BJ_OK (0) or a negative bj_status; the reference panics/asserts instead
 *    (e.g. src/cs/implementations/prover.rs:1425-1438).  bj_last_error() gives the message for the context.
 *  - There is NO CPU fallback: without a usable HIP device bj_ctx_create fails with BJ_ERR_NO_DEVICE.
 */
#ifndef BOOJUM_HIP_H
#define BOOJUM_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum bj_status {
    BJ_OK = 0,
    BJ_ERR_INVALID_ARG = -1,
    BJ_ERR_NO_DEVICE = -2,
    BJ_ERR_HIP = -3,
    BJ_ERR_OOM = -4,
    BJ_ERR_UNSUPPORTED = -5
} bj_status;

/* Opaque per-GPU context: device id, HIP stream, cached twiddle tables, scratch.  Replaces the role of the
 * reference's `Worker` (src/worker/mod.rs:5-87, a rayon pool) + `P::Context` at the call sites. */
typedef struct bj_ctx bj_ctx;

/* 2: bj_gate_desc.wit_stride, bj_comm.all_gather_stream (round 2); 3: op lists in any numbering, run-time compiled gates;
 * 4: bj_proof_config.pow_runner, bj_circuit.table_id_col = BJ_TABLE_ID_AS_VARIABLE (round 5); 5: bj_comm_replay_capture,
 * bj_proof_workspace_bytes, bj_setup_device_bytes; 6: bj_prove_async / bj_proof_wait, the tiled-monomial operators, bj_comm_peer_create (round 6) */
#define BJ_ABI_VERSION 6
int bj_abi_version(void);
int bj_device_count(void);
const char *bj_status_string(int status);

/* The optional BJ_* environment switches (A/B pass plans, test hooks; none changes a result — DESIGN.md §3) are read once, by
 * the first bj_ctx_create of the process; nothing on a proof path calls getenv.  bj_env_reload() reads them again (tests that
 * exercise both sides of a switch in one process); it must not run concurrently with any other call into the library. */
void bj_env_reload(void);
int bj_ctx_create(int device, bj_ctx **out);
void bj_ctx_destroy(bj_ctx *ctx);
/* Use an existing hipStream_t (e.g. PyTorch's current stream) for all subsequent work; NULL = the default stream. */
int bj_ctx_set_stream(bj_ctx *ctx, void *hip_stream);
const char *bj_last_error(const bj_ctx *ctx);
int bj_sync(bj_ctx *ctx);
/* A context keeps the proof workspace (one bump-allocated arena sized by the largest proof so far, NTT scratch, the staging
 * area of bj_prove's host witness) across proofs, the way the reference's prover reuses its `Vec`s inside one `prove_cpu_basic`
 * call (prover.rs:153-168) but no further.  A host that moves from large circuits to small ones gives the memory back with this
 * call; twiddle tables and setups stay.  Not allowed while a proof is running. */
int bj_ctx_release_workspace(bj_ctx *ctx);

/* Device memory helpers for hosts that do not bring their own allocator (the reference threads `A: GoodAllocator`,
 * src/cs/traits/mod.rs:13, through every buffer for exactly this purpose). */
int bj_malloc(bj_ctx *ctx, size_t bytes, void **d_ptr);
int bj_free(bj_ctx *ctx, void *d_ptr);
int bj_memcpy_h2d(bj_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int bj_memcpy_d2h(bj_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int bj_memcpy_d2d(bj_ctx *ctx, void *d_dst, const void *d_src, size_t bytes); /* stream-ordered, then synchronised */
/* Tree hasher used by the bj_merkle_tree_* / bj_fri_prove calls on this context (BJ_HASHER_*, default Poseidon2);
 * bj_prove sets it from its proof config for the duration of the proof. */
int bj_ctx_set_tree_hasher(bj_ctx *ctx, int hasher);

/* HIP-event stopwatch on the context's stream (used by bench.py to time the kernels where they are launched). */
int bj_timer_start(bj_ctx *ctx);
int bj_timer_stop_ms(bj_ctx *ctx, float *ms); /* synchronises on the stop event */

/* ---------------------------------------------------------------------------------------------------------------
 * NTT family.  Replaces PrimeFieldLikeVectorized::{fft_natural_to_bitreversed, ifft_natural_to_natural,
 * precompute_forward/inverse_twiddles_for_fft} (src/field/traits/field_like.rs:139-161), which the reference calls
 * once per polynomial from rayon workers (src/cs/implementations/utils.rs:295-304, 363-379); here one call
 * transforms a whole batch of columns.  Twiddle tables (utils.rs:88-125) are cached inside the context.
 * ------------------------------------------------------------------------------------------------------------- */

/* fft_natural_to_bitreversed (src/fft/mod.rs:398-411): for each column, natural-order coefficients ->
 * evaluations on coset*<w_n> in bit-reversed order.  In place when d_out == d_in. */
int bj_ntt_forward_batch(bj_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, unsigned log_n, unsigned n_cols,
                         size_t col_stride, uint64_t coset);

/* ifft_natural_to_natural (src/fft/mod.rs:464-491): natural-order evaluations on coset*<w_n> -> natural-order
 * monomial coefficients (includes the n^-1 and coset^-i factors).  In place when d_out == d_in. */
int bj_intt_batch(bj_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, unsigned log_n, unsigned n_cols,
                  size_t col_stride, uint64_t coset);

/* transform_monomials_to_lde (src/cs/implementations/utils.rs:311-403): monomials [n_cols][n] (col_stride apart)
 * -> d_out [n_cols][2^log_lde][n]; coset c is evaluated on g*w_{nL}^{bitrev(c)} * <w_n>, bit-reversed.
 * d_out must not alias d_mono. */
int bj_lde_batch(bj_ctx *ctx, const uint64_t *d_mono, size_t col_stride, uint64_t *d_out, unsigned log_n,
                 unsigned n_cols, unsigned log_lde);

/* The same for the cosets [coset_begin, coset_begin + coset_count) only: d_out is [n_cols][coset_count][n].
 * This is the unit of multi-GPU sharding: GPU g owns a contiguous range of cosets of every column,
 * i.e. a contiguous range of Merkle leaves (leaf index = coset*n + i, proof.rs:89-91). */
int bj_lde_cosets_batch(bj_ctx *ctx, const uint64_t *d_mono, size_t col_stride, uint64_t *d_out, unsigned log_n,
                        unsigned n_cols, unsigned log_lde, unsigned coset_begin, unsigned coset_count);

/* transform_raw_storages_to_lde (utils.rs:270-309): natural-order trace columns -> monomials (written back to
 * d_cols) -> LDE in d_out, i.e. bj_intt_batch(coset 1) followed by bj_lde_batch. */
int bj_trace_to_lde_batch(bj_ctx *ctx, uint64_t *d_cols, size_t col_stride, uint64_t *d_out, unsigned log_n,
                          unsigned n_cols, unsigned log_lde);

/* bitreverse_enumeration_inplace (src/fft/mod.rs:41-155), out of place or in place per column. */
int bj_bitreverse_batch(bj_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, unsigned log_n, unsigned n_cols,
                        size_t col_stride);

/* The monomial form between ifft_natural_to_natural (fft/mod.rs:464-491) and transform_monomials_to_lde (utils.rs:311-403) in the
 * layout bj_prove keeps it in.  For 2^22-row traces (the size with a two-pass transform plan) that is a TILED layout — coefficient
 * e = m * 4096 + r * 8 + l (l < 8) at word r * 8192 + (l >> 1) * 2048 + m * 2 + (l & 1) — which the inverse transform's last pass
 * stores directly (the bit reversal and the 1/n factor of the reference's inverse happen in that store: two HBM passes, no
 * bit-reversal pass) and the extension's first pass reads as contiguous 64 KB tiles; every other size keeps natural order and
 * these calls return BJ_ERR_UNSUPPORTED.  Pointwise work on monomials (linear combinations) is layout-blind.
 * bj_monomials_tiled: 1 when bj_prove uses the tiled layout for this trace length (BJ_MONO_TILED=0 / BJ_NTT_TWO_PASS=0 turn it off).
 * bj_intt_batch_tiled: values on the subgroup, natural order -> tiled monomials (main domain, no coset; in place allowed).
 * bj_lde_cosets_batch_tiled: bj_lde_cosets_batch reading tiled monomials; the output is the same as from natural ones.
 * bj_tiled_permute_batch: natural -> tiled (to_tiled != 0) or back, out of place.
 * Columns must start on 16-byte boundaries with an even stride. */
int bj_monomials_tiled(unsigned log_n);
int bj_intt_batch_tiled(bj_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, unsigned log_n, unsigned n_cols, size_t col_stride);
int bj_lde_cosets_batch_tiled(bj_ctx *ctx, const uint64_t *d_mono_tiled, size_t col_stride, uint64_t *d_out, unsigned log_n,
                              unsigned n_cols, unsigned log_lde, unsigned coset_begin, unsigned coset_count);
int bj_tiled_permute_batch(bj_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, unsigned log_n, unsigned n_cols, size_t col_stride,
                           int to_tiled);

/* Reduce every element to its canonical residue (what the reference does on serialisation, goldilocks/mod.rs:98-107). */
int bj_canonicalize(bj_ctx *ctx, uint64_t *d_data, size_t n);

/* Elementwise field operators over n elements (n pairs for BJ_FIELD_EXT2_MUL, second halves at +n): GoldilocksField add / sub /
 * mul / square / inverse (src/field/goldilocks/mod.rs:188-255, 294-360; inverse of 0 is 0 here, the reference returns None)
 * and the F_p^2 product (src/field/traits/field.rs:407-426).  Inputs may be any u64, outputs are canonical.  BJ_FIELD_MUL_LAZY
 * is the same product through the non-canonical ("weak") multiplication the hash / NTT kernels use internally. */
enum { BJ_FIELD_ADD = 0, BJ_FIELD_SUB = 1, BJ_FIELD_MUL = 2, BJ_FIELD_MUL_LAZY = 3, BJ_FIELD_SQUARE = 4, BJ_FIELD_INVERSE = 5,
       BJ_FIELD_EXT2_MUL = 6,
       BJ_FIELD_BUTTERFLY = 7, /* the radix-2 butterfly of serial_ct_ntt (src/fft/mod.rs:659-734) as the NTT kernels run it: d_a = [u | v]
                                * (second half at +n), d_b = w;  out = [u + v*w | u - v*w]; n even */
       BJ_FIELD_ADDSUB = 8,    /* the same with twiddle 1: d_a = [u | v], d_b unused; out = [u + v | u - v] */
       BJ_FIELD_ADD_LAZY = 9, BJ_FIELD_SUB_LAZY = 10, /* sum / difference of the RAW words as weak residues (what the quotient's product
                                * chains run between their weak products), canonicalised only for the store */
       BJ_FIELD_EXT2_MUL_LAZY = 11 /* the F_p^2 product on weak residues: raw words in (second halves at +n) */ };
int bj_field_op_batch(bj_ctx *ctx, int op, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t n);

/* Host-memory convenience for single-polynomial plumbing (what a Rust `impl PrimeFieldLikeVectorized` would call);
 * synchronous, includes the PCIe copies. */
int bj_ntt_forward_host(bj_ctx *ctx, uint64_t *h_inout, unsigned log_n, unsigned n_cols, uint64_t coset);
int bj_intt_host(bj_ctx *ctx, uint64_t *h_inout, unsigned log_n, unsigned n_cols, uint64_t coset);

/* ---------------------------------------------------------------------------------------------------------------
 * Poseidon2 Merkle trees.  Replaces TreeHasher::{hash_into_leaf, hash_into_node} for the Poseidon2 sponge
 * (src/cs/oracle/mod.rs:84-176) and MerkleTreeWithCap::{construct, construct_by_chunking[_from_flat_sources],
 * continue_from_leaf_hashes, get_cap} (src/cs/oracle/merkle_tree.rs:78-460).
 * A tree is stored as all its layers back to back, 4 u64 per digest: layer 0 = num_leaves leaf digests, layer 1 =
 * num_leaves/2, ..., last layer = cap_size digests (the cap).  Size: bj_merkle_tree_digests() * 4 u64.
 * ------------------------------------------------------------------------------------------------------------- */
size_t bj_merkle_tree_digests(size_t num_leaves, size_t cap_size); /* = 2*num_leaves - cap_size */

/* construct: leaf I = hash(col_0[I], col_1[I], ...); columns at d_cols + c*col_stride, each num_leaves long
 * (= lde_factor * n, the [coset][n] layout). */
int bj_merkle_tree_build(bj_ctx *ctx, const uint64_t *d_cols, size_t col_stride, unsigned n_cols, size_t num_leaves,
                         size_t cap_size, uint64_t *d_tree);
/* same, columns given as a HOST array of n_cols DEVICE pointers (leaf order = array order; lets one tree span
 * several column batches, e.g. variables || witness || multiplicities, prover.rs:317-343). */
int bj_merkle_tree_build_ptrs(bj_ctx *ctx, const uint64_t *const *h_col_ptrs, unsigned n_cols, size_t num_leaves,
                              size_t cap_size, uint64_t *d_tree);
/* construct_by_chunking over an F_p^2 codeword: leaf j = hash(c0[jE..(j+1)E) || c1[jE..(j+1)E)), E = 2^log_elems. */
int bj_merkle_tree_build_chunked(bj_ctx *ctx, const uint64_t *d_c0, const uint64_t *d_c1, size_t len,
                                 unsigned log_elems_per_leaf, size_t cap_size, uint64_t *d_tree);
/* continue_from_leaf_hashes: layer 0 of d_tree already holds the leaf digests; fill the upper layers. */
int bj_merkle_tree_nodes(bj_ctx *ctx, uint64_t *d_tree, size_t num_leaves, size_t cap_size);
/* get_cap: copy the cap (cap_size * 4 u64) to the host; synchronous. */
int bj_merkle_tree_cap(bj_ctx *ctx, const uint64_t *d_tree, size_t num_leaves, size_t cap_size, uint64_t *h_cap);
/* get_proof (merkle_tree.rs:462-480): leaf digest + sibling path (depth*4 u64, depth = log2(num_leaves/cap_size))
 * for leaf idx, copied to the host; synchronous. */
int bj_merkle_tree_proof(bj_ctx *ctx, const uint64_t *d_tree, size_t num_leaves, size_t cap_size, size_t idx,
                         uint64_t *h_leaf_digest, uint64_t *h_path);
/* Raw permutation on n_states 12-word states in device memory (testing / transcript offload). */
int bj_poseidon2_permute(bj_ctx *ctx, uint64_t *d_states, size_t n_states);

/* ---------------------------------------------------------------------------------------------------------------
 * FRI.  Replaces fold_multiple / interpolate_* (src/cs/implementations/fri/mod.rs:362-678).
 * ------------------------------------------------------------------------------------------------------------- */
/* One fold by 2 of the bit-reversed F_p^2 array (c0, c1) of length len into (o0, o1) of length len/2:
 *   out[i] = (a + b) + (ch0 + ch1*u) * ((a - b) * roots[i] * coset_inv),  a = in[2i], b = in[2i+1].
 * roots = inverse bit-reversed twiddles of the initial LDE domain of size 2^log_full (cached in the context). */
int bj_fri_fold(bj_ctx *ctx, const uint64_t *d_c0, const uint64_t *d_c1, size_t len, uint64_t *d_o0, uint64_t *d_o1,
                unsigned log_full, uint64_t coset_inv, uint64_t ch0, uint64_t ch1);

/* fold by 2^k (k = 1..3) in ONE launch — one step of the folding schedule; alpha and coset_inv are squared between
 * the inner folds inside the kernel exactly as interpolate_flattened_cosets does (fri/mod.rs:587-678). */
int bj_fri_fold_step(bj_ctx *ctx, const uint64_t *d_c0, const uint64_t *d_c1, size_t len, unsigned k, uint64_t *d_o0,
                     uint64_t *d_o1, unsigned log_full, uint64_t coset_inv, uint64_t ch0, uint64_t ch1);

/* ---------------------------------------------------------------------------------------------------------------
 * Openings (prover round 4 and the DEEP part of round 5, prover.rs:1501-2067).
 * ------------------------------------------------------------------------------------------------------------- */
/* precompute_for_barycentric_evaluation_in_extension (src/cs/implementations/utils.rs:907-1021): weights for
 * evaluating at at = (at2[0] + at2[1] u) from the 2^log_n values on coset*<w_n> given in BIT-REVERSED order
 * (i.e. coset 0 of an LDE column when coset = 7).  d_w0/d_w1 receive 2^log_n values each. */
int bj_barycentric_weights(bj_ctx *ctx, unsigned log_n, uint64_t coset, const uint64_t *at2, uint64_t *d_w0,
                           uint64_t *d_w1);
/* barycentric_evaluate_base_at_extension_for_bitreversed_parallel (utils.rs:1085-1158) for a batch of base-field
 * columns: h_out[2c], h_out[2c+1] = sum_i col_c[i] * w[i].  For an F_p^2 polynomial stored as columns (c0, c1) the
 * value is (E(c0).0 + 7*E(c1).1, E(c0).1 + E(c1).0) (= the _extension_at_extension_ variant, utils.rs:1160-1242).
 * h_col_ptrs: host array of device pointers.  Synchronous (results are needed by the transcript). */
int bj_barycentric_eval_batch(bj_ctx *ctx, const uint64_t *const *h_col_ptrs, unsigned n_cols, unsigned log_n,
                              const uint64_t *d_w0, const uint64_t *d_w1, uint64_t *h_out);
/* quotening_operation_in_extension (src/cs/implementations/prover.rs:2523-2706):
 *   dst[I] (+)= [ sum_k ch_k * (f_k[I] - v_k) ] / (x_I - at),  x_I = g*w_{nL}^{bitrev(I)}, I < 2^(log_n+log_lde).
 * Source k is the LDE column h_src_c0[k] (and h_src_c1[k] for an F_p^2 polynomial; NULL entry or NULL array = base
 * field).  h_values / h_challenges are [n_src][2].  accumulate = 0 overwrites dst, otherwise adds to it. */
int bj_deep_quotient_accumulate(bj_ctx *ctx, const uint64_t *const *h_src_c0, const uint64_t *const *h_src_c1,
                                size_t n_src, const uint64_t *h_values, const uint64_t *h_challenges,
                                const uint64_t *at2, unsigned log_n, unsigned log_lde, uint64_t *d_dst_c0,
                                uint64_t *d_dst_c1, int accumulate);

/* ---------------------------------------------------------------------------------------------------------------
 * Fiat–Shamir transcript (host side, tiny data, order-critical).  Replaces `Transcript` impls
 * (src/cs/implementations/transcript.rs:7-42): BJ_TRANSCRIPT_POSEIDON2 = GoldilocksPoisedon2Transcript
 * (transcript.rs:144-151: algebraic sponge, rate 8, overwrite mode, "1"-padding).  query_index = BoolsBuffer::get_bits
 * (transcript.rs:369-417) + the inner/coset split of prover.rs:2161-2182; returns coset*n + inner.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct bj_transcript bj_transcript;
#define BJ_TRANSCRIPT_POSEIDON2 1
/* GoldilocksPoisedonTranscript (transcript.rs:133-141): the same sponge over the Poseidon (v1) permutation, which is what
 * the SHA-256 bench script pairs with the Poseidon2 tree hasher (gadgets/sha256/mod.rs:289-293).  The reference holds no
 * known-answer vector for that permutation. */
#define BJ_TRANSCRIPT_POSEIDON 2
/* Blake2sTranscript (transcript.rs:155-262): byte transcript over Blake2s-256 (RFC 7693), paired with the Blake2s tree
 * hasher in the non-recursive configuration (gadgets/sha256/mod.rs:265-270).  Caps are absorbed as raw digest bytes. */
#define BJ_TRANSCRIPT_BLAKE2S 3
#define BJ_TRANSCRIPT_KECCAK256 4 /* Keccak256Transcript (transcript.rs:264-372): the same byte transcript over Keccak-256 */

/* Tree hashers (TreeHasher impls, src/cs/oracle/mod.rs:114-245).  Digests are 32 bytes = four u64 words either way. */
#define BJ_HASHER_POSEIDON2 1 /* GoldilocksPoseidon2Sponge<AbsorptionModeOverwrite>: four canonical field elements */
#define BJ_HASHER_BLAKE2S 2   /* blake2::Blake2s256: the 32 digest bytes, little-endian packed into the four words */
#define BJ_HASHER_KECCAK256 3 /* sha3::Keccak256 (original 0x01 padding, oracle/mod.rs:247-312): digest bytes as above */
/* bj_proof_config.pow_runner: the PoWRunner implementations of src/cs/implementations/pow.rs */
#define BJ_POW_BLAKE2S256 1   /* pow.rs:50-133 */
#define BJ_POW_KECCAK256 2    /* pow.rs:139-230 (original Keccak padding, as the tree hasher) */
int bj_transcript_create(int kind, bj_transcript **out);
void bj_transcript_destroy(bj_transcript *t);
int bj_transcript_absorb(bj_transcript *t, const uint64_t *els, size_t n);
/* witness_merkle_tree_cap: digests as 4 u64 words each; field elements for the algebraic transcripts, raw bytes for Blake2s */
int bj_transcript_absorb_cap(bj_transcript *t, const uint64_t *digest_words, size_t n_words);
int bj_transcript_challenge(bj_transcript *t, uint64_t *out);
int bj_transcript_query_index(bj_transcript *t, unsigned log_n, unsigned log_lde, uint64_t *out_index);

/* compute_fri_schedule (src/cs/implementations/prover.rs:2281-2372); `schedule` must hold >= 32 entries. */
int bj_fri_schedule(uint32_t security_bits, size_t cap_size, uint32_t pow_bits, uint32_t rate_log2,
                    uint32_t initial_degree_log2, uint32_t *new_pow_bits, size_t *num_queries, uint32_t *schedule,
                    size_t *schedule_len, size_t *final_degree);

/* do_fri (src/cs/implementations/fri/mod.rs:49-358): commit phase over the F_p^2 codeword (d_c0, d_c1) of length
 * 2^(log_n+log_lde) laid out [coset][n] bit-reversed.  For every schedule step: Poseidon2 oracle with 2^k values per
 * leaf -> cap into the transcript -> challenge (c0, c1) -> fold by 2^k.  Finally bit-reverse + iNTT of the last
 * layer, final monomials into the transcript.  The returned object keeps oracles and leaf sources in HBM for the
 * query phase; the caller's codeword must stay alive until bj_fri_destroy.  Returns BJ_ERR_INVALID_ARG if the
 * codeword is not of low degree (the reference asserts, fri/mod.rs:327-336). */
typedef struct bj_fri bj_fri;
int bj_fri_prove(bj_ctx *ctx, const uint64_t *d_c0, const uint64_t *d_c1, unsigned log_n, unsigned log_lde,
                 const uint32_t *schedule, size_t schedule_len, size_t cap_size, bj_transcript *transcript,
                 bj_fri **out);
void bj_fri_destroy(bj_fri *f);
size_t bj_fri_num_oracles(const bj_fri *f);
size_t bj_fri_final_degree(const bj_fri *f);
int bj_fri_cap(const bj_fri *f, size_t oracle, uint64_t *h_cap);          /* cap_size*4 u64 */
int bj_fri_challenge(const bj_fri *f, size_t oracle, uint64_t *h_ch2);    /* the (c0,c1) drawn after that cap */
int bj_fri_final_monomials(const bj_fri *f, uint64_t *h_c0, uint64_t *h_c1); /* final_degree u64 each */
/* OracleQuery::construct for FRI oracles (proof.rs:65-100, fri/mod.rs:829-895): `index` is the flat index into that
 * oracle's own source array; returns 2*2^k leaf elements (c0 run then c1 run) + the path (depth*4 u64). */
int bj_fri_query(bj_ctx *ctx, const bj_fri *f, size_t oracle, size_t index, uint64_t *h_leaf_elements,
                 uint64_t *h_path);

/* ---------------------------------------------------------------------------------------------------------------
 * Whole prover (seam S1).  Replaces CSReferenceAssembly::prove_cpu_basic (src/cs/implementations/prover.rs:153-168)
 * and the prover-side part of get_full_setup (src/cs/implementations/setup.rs:1273-1300).  Circuit class: gates over
 * general-purpose columns selected by a selector tree (the SHA-256 bench's four hand-written, any other evaluator as its
 * gpu_synthesizer op list), gates over specialized columns, specialized lookups with a shared constant table id, witness
 * (non-copiable) columns; tree hasher Poseidon2 / Blake2s / Keccak-256 with the matching transcript (and the bench script's
 * Poseidon2 tree + Poseidon transcript pairing); Blake2s proof of work up to 32 bits.
 * ------------------------------------------------------------------------------------------------------------- */
typedef enum bj_gate_kind {
    BJ_GATE_CONSTANT_ALLOCATOR = 1, /* a - c                      src/cs/gates/constant_allocator.rs:107-126          */
    BJ_GATE_FMA_NO_CONSTANT = 2,    /* q*a*b + l*c - d            src/cs/gates/fma_gate_without_constant.rs:96-126    */
    BJ_GATE_REDUCTION4 = 3,         /* sum_i c_i*v_i - r          src/cs/gates/reduction_gate.rs:103-126              */
    BJ_GATE_NOP = 4,                /* no terms                   src/cs/gates/nop_gate.rs:41                         */
    BJ_GATE_PROGRAM = 5,            /* any evaluator, given as the op list of seam S3 (bj_gate_program below)          */
    BJ_GATE_POSEIDON2_FLATTENED = 6 /* Poseidon2FlattenedGate<8,12,4>, 130 variables, 118 terms, one repetition per row:
                                     * src/cs/gates/poseidon2.rs:165-410 (the gate of the recursion circuits), hand-written */
} bj_gate_kind;

/* Seam S3 — a gate as the reference's gpu_synthesizer describes it (GPUDataCapture::from_evaluator,
 * src/gpu_synthesizer/mod.rs:113-133, 354-444): `relations` are executed in order, each writes one temporary;
 * operands are variable columns (relative to the repetition), constant columns (relative to the end of the selector
 * path, plus the repetition's constant offset), temporaries or field constants; `writes` are the quotient terms of one
 * repetition in push order (GPUPolyDestination::push_evaluation_result).  Witness columns (Index::WitnessPoly) are relative to
 * the repetition's witness offset (bj_gate_desc.wit_stride). */
typedef enum bj_gate_op { /* Relation<F>, gpu_synthesizer/mod.rs:123-133 */
    BJ_OP_ADD = 1, BJ_OP_DOUBLE = 2, BJ_OP_SUB = 3, BJ_OP_NEGATE = 4, BJ_OP_MUL = 5, BJ_OP_SQUARE = 6, BJ_OP_INVERSE = 7
} bj_gate_op;
typedef enum bj_index_kind { /* Index<F>, gpu_synthesizer/mod.rs:113-121 */
    BJ_IDX_VARIABLE_POLY = 0, BJ_IDX_WITNESS_POLY = 1, BJ_IDX_CONSTANT_POLY = 2, BJ_IDX_TEMPORARY = 3, BJ_IDX_CONSTANT_VALUE = 4
} bj_index_kind;
typedef struct bj_gate_index {
    uint32_t kind;  /* bj_index_kind */
    uint32_t index; /* column / temporary number; for BJ_IDX_CONSTANT_VALUE an index into bj_gate_program::values */
} bj_gate_index;
typedef struct bj_gate_relation {
    uint32_t op;  /* bj_gate_op */
    uint32_t dst; /* temporary written */
    bj_gate_index a, b; /* b is ignored by the unary ops */
} bj_gate_relation;
typedef struct bj_gate_program {
    const bj_gate_relation *relations;
    uint32_t num_relations;
    const uint64_t *values; /* field constants referenced by BJ_IDX_CONSTANT_VALUE */
    uint32_t num_values;
    const bj_gate_index *writes; /* num_terms entries */
    uint32_t num_writes;
    uint32_t num_temporaries; /* every relations[i].dst < num_temporaries; any numbering (a temporary may be written once, as the
                               * reference does, or reused): slots are assigned by the library */
} bj_gate_program;

typedef struct bj_gate_desc {
    int kind;                 /* bj_gate_kind */
    unsigned path_len;        /* selector path from TreeNode::output_placement (setup.rs:1455-1483) */
    unsigned char path[8];    /* 1: multiply by constant column i, 0: by (1 - constant column i)  (prover.rs:2775-2916) */
    unsigned num_repetitions; /* num_repetitions_in_geometry */
    unsigned var_stride;      /* per_chunk_offset.variables_offset */
    unsigned wit_stride;      /* per_chunk_offset.witnesses_offset (0 for evaluators that read no witness column) */
    unsigned const_stride;    /* per_chunk_offset.constants_offset */
    unsigned num_terms;       /* quotient terms per repetition (0 for markers) */
    const bj_gate_program *program; /* BJ_GATE_PROGRAM only, NULL otherwise */
} bj_gate_desc;

/* What the library does with an op list, in this order: (1) it is brought into canonical form (era_boojum_amd/csrc/gate_canon.h:
 * DAG with common subexpressions merged, slots by live range, a structural fingerprint that does not depend on the numbering
 * of temporaries or on the order of independent relations) — so a GPUDataCapture goes in as the reference records it, one
 * fresh temporary per operation from its process-wide counter (gpu_synthesizer/mod.rs:210-352), renumbered densely or not;
 * (2) if the fingerprint is that of an evaluator the library was built with (every evaluator of src/cs/gates/, traced call by
 * call in era_boojum_amd/gate_program.py), its build-time straight-line kernel runs — the reference's capture of
 * Poseidon2FlattenedGate selects the hand-written evaluator; (3) otherwise a straight-line kernel is compiled from the op
 * list at bj_setup_create with hiprtc and cached per process (and on disk when BJ_GATE_JIT_CACHE names a directory);
 * (4) without hiprtc, or with BJ_GATE_NO_JIT set, an interpreter kernel runs the canonical schedule.  Same terms every way. */

/* The fingerprint is a NON-CRYPTOGRAPHIC 128-bit mix of the canonical DAG (csrc/gate_canon.cpp): it selects kernels for op lists
 * that come from a trusted host (the reference's own gpu_synthesizer), it is not a commitment to them.  A hit on a build-time
 * kernel is believed only together with the program's structural summary (operations, slots, terms, column extents) recorded
 * at build time, and the run-time compiler's caches are keyed by the emitted source text as well; a program that collides on
 * purpose can at worst select the interpreter for itself. */
/* 1 if the library carries a build-time kernel for this program's function (2 above), 0 otherwise.  Host-only, no GPU. */
int bj_gate_program_generated(const bj_gate_program *program);

/* Canonical form of a program: fp[2] = structural fingerprint, *num_slots = values live at once in the canonical schedule,
 * *num_ops = operations after merging, extents3 = {variable, constant, witness} columns one repetition reads.  Any out pointer
 * may be NULL.  BJ_ERR_INVALID_ARG for a malformed list (use of an unwritten temporary, bad operand).  Host-only. */
int bj_gate_program_canonical_info(const bj_gate_program *program, uint64_t fp[2], uint32_t *num_slots, uint32_t *num_ops,
                                   uint32_t *extents3);
/* The straight-line statements of the canonical program (what both the build-time generator and the run-time compiler emit),
 * NUL-terminated into out[cap]; returns the size needed including the terminator (call with out = NULL to ask), 0 on error. */
size_t bj_gate_program_emit_body(const bj_gate_program *program, char *out, size_t cap);
/* The complete HIP source the run-time compiler is given for this program; same calling convention. */
size_t bj_gate_program_jit_source(const bj_gate_program *program, char *out, size_t cap);
/* Compile that source for `arch` (NULL: "gfx950") without loading it: BJ_OK, BJ_ERR_UNSUPPORTED when hiprtc is not installed,
 * BJ_ERR_INVALID_ARG with the compiler's log in log[log_cap].  Host-only (works without a GPU). */
int bj_gate_program_jit_compile_check(const bj_gate_program *program, const char *arch, char *log, size_t log_cap);
/* Number of gate kernels this process obtained at run time (compiled or read from BJ_GATE_JIT_CACHE); a one-line status or
 * the reason of the last failure into out[cap]. */
int bj_gate_jit_status(char *out, size_t cap);

/* Evaluate a gate program at n_points points (stand-alone, for parity tests of S3): d_terms[(rep*num_writes + t)*n_points + i]
 * = term t of repetition rep at point i.  d_vars / d_consts: columns with the given strides, constants WITHOUT a selector
 * prefix. */
int bj_gate_program_eval(bj_ctx *ctx, const bj_gate_program *program, const uint64_t *d_vars, size_t var_stride,
                         const uint64_t *d_consts, size_t const_stride, unsigned num_repetitions, unsigned rep_var_stride,
                         unsigned rep_const_stride, size_t n_points, uint64_t *d_terms);

/* ---------------------------------------------------------------------------------------------------------------
 * Seam S2 for the second round and the quotient terms: the kernels bj_prove runs, on caller-provided device columns.
 * F_p^2 challenges are two u64 (c0, c1) in host memory; F_p^2 columns are (c0, c1) pairs of base columns.
 * ------------------------------------------------------------------------------------------------------------- */

/* compute_partial_products_in_extension + shifted_grand_product_in_extension (src/cs/implementations/copy_permutation.rs:
 * 649-830, 425-510; per-row rationals :114-248): variables / sigmas [num_vars][n] in natural row order over the main domain,
 * columns taken in chunks of `chunk` (= quotient degree).  Out: d_z [2][n] (z(1) = 1) and d_partials [n_chunks-1][2][n]. */
int bj_copy_perm_stage2(bj_ctx *ctx, const uint64_t *d_vars, size_t var_stride, const uint64_t *d_sigmas, size_t sig_stride,
                        const uint64_t *h_non_residues, unsigned num_vars, unsigned chunk, unsigned log_n,
                        const uint64_t *h_beta, const uint64_t *h_gamma, uint64_t *d_z, uint64_t *d_partials);
/* compute_lookup_poly_pairs_specialized (src/cs/implementations/lookup_argument_in_ext.rs:320-700): A_i = 1 / (beta +
 * sum_j gamma^j col_ij + gamma^width table_id), B = multiplicity / (beta + sum_j gamma^j table_j) per row.
 * d_lookup_vars [reps*width][n], d_tables [width+1][n]; out d_A [reps][2][n], d_B [2][n].
 * d_table_id == NULL selects LookupParameters::UseSpecializedColumnsWithTableIdAsVariable (:354-366): d_lookup_vars is
 * [reps*(width+1)][n], the last column of every sub-argument carries its table id and no constant column takes part. */
int bj_lookup_polys(bj_ctx *ctx, const uint64_t *d_lookup_vars, size_t var_stride, const uint64_t *d_table_id,
                    const uint64_t *d_tables, size_t table_stride, const uint64_t *d_multiplicities, unsigned reps, unsigned width,
                    unsigned log_n, const uint64_t *h_beta, const uint64_t *h_gamma, uint64_t *d_A, uint64_t *d_B);
/* Gate terms of the quotient numerator at num_points LDE points (prover.rs:1031-1080 with buffering_source.rs:133-362 and
 * the selectors of prover.rs:2775-2916): out = sum_g selector_g * sum_t alpha_t * term_t for the hand-written evaluators
 * (kinds 1..4; op lists are evaluated by bj_gate_program_eval).  h_alphas: one F_p^2 power per (gate, repetition, term) in
 * evaluator order.  OVERWRITES d_out0 / d_out1. */
int bj_quotient_gates(bj_ctx *ctx, const uint64_t *d_vars, size_t var_stride, unsigned num_gp_vars, const uint64_t *d_consts,
                      size_t const_stride, unsigned num_constant_cols, const bj_gate_desc *gates, unsigned num_gates,
                      const uint64_t *h_alphas, size_t num_points, uint64_t *d_out0, uint64_t *d_out1);
/* compute_quotient_terms_for_lookup_specialized (lookup_argument_in_ext.rs:949-1319): ADDS sum_i alpha_i * (A_i * denom_i - 1)
 * + alpha_reps * (B * denom_table - multiplicity) to d_out.  h_alphas: reps + 1 powers.  d_table_id == NULL: table id as the
 * (width+1)-th variable column of every sub-argument, as for bj_lookup_polys. */
int bj_quotient_lookup(bj_ctx *ctx, const uint64_t *d_lookup_vars, size_t var_stride, const uint64_t *d_table_id,
                       const uint64_t *d_tables, size_t table_stride, const uint64_t *d_multiplicities, const uint64_t *d_A,
                       const uint64_t *d_B, size_t stage2_stride, unsigned reps, unsigned width, const uint64_t *h_beta,
                       const uint64_t *h_gamma, const uint64_t *h_alphas, size_t num_points, uint64_t *d_out0, uint64_t *d_out1);
/* (z - 1) * L1 (prover.rs:1189-1227), the copy-permutation chain compute_quotient_terms_in_extension
 * (copy_permutation.rs:1000-1249) and divide_by_vanishing_for_bitreversed_coset_enumeration (utils.rs:770-817): ADDS the
 * terms to d_out and then multiplies by 1 / (x^n - 1).  Points are the flat LDE indices first_point .. first_point +
 * num_points (first_point at a coset boundary, bit-reversed inside a coset; num_points need not be whole cosets: a rank of a
 * sharded proof evaluates the first q n / world points of its range, bj_combine_residues below); d_stage2 = z, partial products as (c0, c1) columns;
 * h_alphas: 1 (L1 term) + n_chunks powers. */
int bj_quotient_copy_perm(bj_ctx *ctx, const uint64_t *d_vars, size_t var_stride, const uint64_t *d_sigmas, size_t sig_stride,
                          const uint64_t *d_stage2, size_t stage2_stride, const uint64_t *h_non_residues, unsigned num_vars,
                          unsigned chunk, unsigned log_n, unsigned log_lde, const uint64_t *h_beta, const uint64_t *h_gamma,
                          const uint64_t *h_alphas, size_t num_points, size_t first_point, uint64_t *d_out0, uint64_t *d_out1);
/* Sharded quotient (the reference's single-host flow is prover.rs:1386-1482: evaluations on q cosets, one size-q n
 * inverse transform): every rank evaluates the terms on q n / world points of its OWN cosets — first_point .. first_point +
 * num_points of the three calls above may be a fraction of a coset — and inverse-transforms them (bj_intt_batch with the
 * coset shift x_{first_point}) to R_i = T mod (x^E - a_i), E = residue_len, a_i = x_{first_point}^E.  This call solves the
 * world x world Vandermonde system coefficient by coefficient: d_residues = [world][num_cols][E] (the all-gathered blocks),
 * h_moduli[i] = a_i (pairwise distinct), d_out = [num_cols][world * E] = the monomial coefficients of T (canonical). */
int bj_combine_residues(bj_ctx *ctx, const uint64_t *d_residues, unsigned world, size_t residue_len, unsigned num_cols,
                        const uint64_t *h_moduli, uint64_t *d_out);

#define BJ_TABLE_ID_AS_VARIABLE 0xFFFFFFFFu   /* bj_circuit.table_id_col: the table id is a variable column, see below */

typedef struct bj_circuit {
    unsigned log_n;              /* trace length 2^log_n */
    unsigned num_vars;           /* all variable columns: general purpose first, then the specialized lookup columns, then the
                                  * columns of the gates over specialized columns */
    unsigned num_gp_vars;        /* CSGeometry::num_columns_under_copy_permutation */
    unsigned num_witness_cols;   /* CSGeometry::num_witness_columns: non-copiable columns (WitnessSet::witness, witness.rs:25).  They
                                  * are committed with the variables (leaf = variables || witness || multiplicities, prover.rs:317-347),
                                  * opened after them, and readable only by op-list gates (BJ_IDX_WITNESS_POLY).  bj_prove / bj_prove_dev
                                  * take them right behind the variable columns: [num_vars + num_witness_cols][n] */
    unsigned num_constant_cols;  /* selector/gate constants + (lookups) the table-id column */
    unsigned lookup_width;       /* LookupParameters::UseSpecializedColumnsWithTableIdAs{Constant,Variable} { width, .. } (cs/mod.rs:237-246) */
    unsigned lookup_reps;        /* num_repetitions; 0 = no lookup argument */
    unsigned table_id_col;       /* ..AsConstant { share_table_id: true }: vk.fixed_parameters.table_ids_column_idxes[0];
                                  * ..AsVariable (table_ids_column_idxes is empty, setup.rs:970-971): BJ_TABLE_ID_AS_VARIABLE — every
                                  * sub-argument then owns lookup_width + 1 variable columns, the last one holding the table id of
                                  * the row (lookup_argument_in_ext.rs:354-366, 949-1000; verifier.rs:1402-1464), num_vars counts them
                                  * and there is no table-id column among the constants */
    unsigned quotient_degree;    /* vk.fixed_parameters.quotient_degree (power of two) */
    unsigned num_gates;          /* evaluators over general purpose columns, in evaluator order (<= 16) */
    const bj_gate_desc *gates;
    const uint64_t *non_residues;     /* num_vars entries: non_residues_for_copy_permutation (copy_permutation.rs:512-523) */
    unsigned num_public_inputs;
    const unsigned *public_input_cols; /* vk.fixed_parameters.public_inputs_locations */
    const unsigned *public_input_rows;
    /* Gates placed with GatePlacementStrategy::UseSpecializedColumns (gate.rs; evaluator.rs:190-236, prover.rs:635-800):
     * no selector, applied on every row to their own variable columns, which follow the lookup columns in declaration
     * order (num_repetitions * var_stride columns each; num_vars counts them).  Each is an op list (kind BJ_GATE_PROGRAM,
     * path_len 0).  One that reads constants (share_constants = false: every repetition has its own principal_width.num_constants
     * columns, const_stride = per_repetition_offset.constants_offset, evaluator_data.rs:196-238) finds them in the LAST constant
     * columns — behind the general-purpose gates' constants and the table-id column, which is the first special-purpose
     * constant (setup.rs:963-1010) — num_repetitions * const_stride columns per gate in declaration order (prover.rs:748-772);
     * num_constant_cols counts them.  Constants shared by the repetitions are refused (the reference hands such an evaluator an
     * empty constant range).  Their quotient terms sit between the lookup terms and the general-purpose gates' terms in the
     * order of the alpha powers (prover.rs:599-625).  0 / NULL when there are none. */
    unsigned num_specialized_gates;
    const bj_gate_desc *specialized_gates;
} bj_circuit;

typedef struct bj_proof_config { /* ProofConfig, prover.rs:55-73 */
    unsigned fri_lde_factor;
    unsigned cap_size;
    unsigned security_level;
    unsigned pow_bits; /* proof of work (pow.rs), <= 32; 0 = off as in the benches (NoPow).  The runner is pow_runner below */
    unsigned transcript;  /* 0 or BJ_TRANSCRIPT_POSEIDON2 (default), BJ_TRANSCRIPT_POSEIDON, BJ_TRANSCRIPT_BLAKE2S, _KECCAK256 */
    unsigned tree_hasher; /* 0 or BJ_HASHER_POSEIDON2 (default, with an algebraic transcript), BJ_HASHER_BLAKE2S /
                           * BJ_HASHER_KECCAK256 (with a byte transcript): the transcript's CompatibleCap must be the hasher's Output */
    unsigned pow_runner;  /* the POW type parameter of prove_cpu_basic (prover.rs:153-168; trait PoWRunner, pow.rs:6-31), independent of
                           * the transcript and the tree hasher as in the reference: 0 or BJ_POW_BLAKE2S256 (impl PoWRunner for
                           * Blake2s256, pow.rs:50-133), BJ_POW_KECCAK256 (impl PoWRunner for Keccak256, pow.rs:139-230).  Both search
                           * the smallest nonce whose H(seed || le64(nonce)) starts with pow_bits trailing zero bits; read only when
                           * pow_bits != 0 */
} bj_proof_config;

typedef struct bj_setup bj_setup; /* device-resident SetupStorage + setup Merkle tree + VK cap; reusable across proofs */
typedef struct bj_proof bj_proof;

/* h_sigmas [num_vars][n], h_constants [num_constant_cols][n], h_tables [lookup_width+1][n] (NULL without lookups):
 * natural-order values over the main domain (SetupBaseStorage, polynomial_storage.rs:48-75). */
int bj_setup_create(bj_ctx *ctx, const bj_circuit *circuit, const uint64_t *h_sigmas, const uint64_t *h_constants,
                    const uint64_t *h_tables, const bj_proof_config *config, bj_setup **out);
/* ---- one proof across several GPUs ----------------------------------------------------------------
 * One process per GPU.  GPU `rank` of `world` owns the cosets [rank*L/world, (rank+1)*L/world) of every LDE'd column
 * (L = fri_lde_factor), which is the contiguous range [rank*N/world, (rank+1)*N/world) of Merkle leaves of every oracle
 * (leaf index = coset*n + i, proof.rs:89-91), hashes its own subtree down to cap_size/world cap nodes, and the ranks
 * exchange only: cap fragments, the quotient evaluations (when a rank owns fewer than quotient_degree cosets), the first
 * folded FRI layer and the query openings.  Main-domain work (iNTTs, copy-permutation and lookup polynomials, openings at
 * z from the rank's own first coset) is replicated, so every rank ends with the same transcript state and the SAME proof,
 * bit for bit the one a single GPU produces.  The all-gather is the library's own (bj_comm_rccl_create below: ncclAllGather
 * of RCCL on the proof's stream, librccl dlopen'ed) or one the host hands in through bj_comm (era_boojum_amd/binding.py wraps
 * torch.distributed that way: gloo in the tests).  Requirements: world a power of two <= 8 (one node: the W x W residue
 * combination of the quotient keeps its inverse Vandermonde matrix, W^2 <= 64 words, in kernel arguments), world | fri_lde_factor,
 * world | cap_size, quotient_degree <= fri_lde_factor. */
typedef struct bj_comm {
    unsigned rank, world;
    /* every rank contributes `bytes` at d_send; on return d_recv holds world*bytes, rank-major.  Called with the
     * context's stream idle; the data must be visible to that stream when it returns.  Non-zero return = failure. */
    int (*all_gather)(void *user, const void *d_send, void *d_recv, size_t bytes);
    void *user;
    /* optional, preferred when set: the same exchange ENQUEUED on the given hipStream_t (the context's stream) and returning at
     * once — stream order replaces both synchronisations.  bj_comm_rccl_create sets it. */
    int (*all_gather_stream)(void *user, const void *d_send, void *d_recv, size_t bytes, void *hip_stream);
} bj_comm;
/* 1 if librccl could be loaded in this process (dlopen; BJ_RCCL_LIB overrides the name), 0 otherwise.  bj_comm_rccl_create is
 * a collective: a host agrees on this flag across its processes FIRST (any channel), so that no rank enters the collective
 * while another one cannot. */
int bj_rccl_available(void);
/* The in-library transport: RCCL (ncclAllGather over xGMI) on the context's stream, directly on the prover's buffers; librccl
 * is loaded at run time (BJ_RCCL_LIB overrides the search).  Rank 0 makes the id and hands its 128 bytes to the other ranks by
 * whatever channel the host has; every rank then calls bj_comm_rccl_create (collective) with its device current, and passes
 * the filled bj_comm to bj_setup_create_sharded.  A Rust / C host needs nothing else to shard a proof over the GPUs of a node. */
#define BJ_RCCL_UNIQUE_ID_BYTES 128
int bj_rccl_unique_id(void *out_id);
int bj_comm_rccl_create(bj_ctx *ctx, const void *unique_id, unsigned rank, unsigned world, bj_comm *out);
void bj_comm_rccl_destroy(bj_comm *comm);
int bj_comm_rccl_stats(const bj_comm *comm, size_t *calls, size_t *bytes_received); /* collectives issued, bytes received */
/* Full-mesh peer transport for the BULK exchanges ("direct full-mesh peer copies so all 7 xGMI links carry
 * traffic"; the reference has no multi-device path, the exchanged data are those of proof.rs:89-91 / merkle_tree.rs:112-157): wraps
 * `base` (bj_comm_rccl_create or a host callback) and serves every exchange of at least bulk_threshold_bytes per rank (0: 1 MiB) —
 * the quotient residues, the first folded FRI layer, the DEEP numerator slices — by copying this rank's contribution straight into
 * slot `rank` of every peer's MAILBOX (a double-buffered receive buffer owned by the transport, exported once with
 * hipIpcGetMemHandle and mapped by the peers with hipIpcOpenMemHandle; world concurrent copies: one per link) instead of a ring that
 * is bound by ONE link, followed by one device copy from the mailbox into d_recv; smaller exchanges go to `base` unchanged.
 * `exchange` is the host's control channel: a BLOCKING all-gather of `bytes` host bytes per rank, rank-major into h_recv
 * (MPI_Allgather, a TCP store, torch.distributed); it carries the IPC handles and a vote once per exchange size and then serves as
 * the completion barrier, one call per bulk exchange.  When any rank cannot export or map a mailbox all ranks agree to use `base`
 * for that size (bj_comm_peer_stats: fallbacks).  Same bytes in the same slots: proofs do not change.  Selectable beside RCCL, never
 * the default; its author had one GPU (the tests run the ranks as processes sharing the device) — no link-level timing exists.
 * `base` must outlive the returned comm. */
typedef int (*bj_host_exchange_fn)(void *user, const void *h_send, void *h_recv, size_t bytes);
int bj_comm_peer_create(bj_ctx *ctx, const bj_comm *base, bj_host_exchange_fn exchange, void *exchange_user, size_t bulk_threshold_bytes,
                        bj_comm *out);
void bj_comm_peer_destroy(bj_comm *comm);
int bj_comm_peer_stats(const bj_comm *comm, size_t *bulk_calls, size_t *bulk_bytes_received, size_t *small_calls, size_t *fallbacks);
int bj_setup_create_sharded(bj_ctx *ctx, const bj_circuit *circuit, const uint64_t *h_sigmas, const uint64_t *h_constants,
                            const uint64_t *h_tables, const bj_proof_config *config, const bj_comm *comm, bj_setup **out);
/* Replaces the transport of a sharded setup (a host that re-creates its communicator, or switches between its own callback and
 * the in-library one): same rank and world as the setup was created with, nothing else changes.  Not while a proof runs. */
int bj_setup_set_comm(bj_setup *s, const bj_comm *comm);
/* Recorded-peer transport (measurement): rank `rank` of a `world`-rank proof runs ALONE on its GPU; the k-th
 * all-gather is served by a device-to-device copy, on the proof's stream, of d_gathered[k] — the world * bytes that collective
 * produced in a real run of the same proof (every rank of a sharded proof receives the same bytes; proofs of one witness are
 * deterministic).  The first n_setup buffers are consumed once (the collectives of bj_setup_create_sharded), the next
 * n_per_proof cyclically by every proof.  The prover cannot tell it from RCCL and the proof is bit for bit the single-GPU one;
 * what it times is one rank's critical path — kernels, launches, host round trips — with the link time replaced by an HBM
 * copy (bench.py --replay-world).  verify != 0: every contribution of this rank is compared with the slice the recording holds
 * for it (synchronising; bj_comm_replay_stats counts the mismatches).  The buffers stay the caller's and must outlive the comm. */
int bj_comm_replay_create(bj_ctx *ctx, unsigned rank, unsigned world, const void *const *d_gathered, const size_t *gathered_bytes,
                          size_t n_setup, size_t n_per_proof, int verify, bj_comm *out);
void bj_comm_replay_destroy(bj_comm *comm);
int bj_comm_replay_stats(const bj_comm *comm, size_t *calls, size_t *bytes_received, size_t *mismatches);
/* Making the recording with ONE rank on the device at a time: a replay transport that holds the first k collectives and has a
 * capture buffer does not serve collective k — it copies this rank's contribution to it into d_capture and fails the call,
 * which ends the proof (bj_prove_dev returns BJ_ERR_HIP; bj_comm_replay_captured then reports the contribution's size, 0 when
 * the proof ended without reaching an unrecorded collective).  The `world` contributions, rank-major, ARE the gathered buffer
 * of collective k: every rank receives the same bytes and proofs of one witness are deterministic.  A 2^23-row proof over eight
 * ranks is recorded this way on one 288 GB device, which cannot hold the eight ranks' workspaces side by side
 * (era_boojum_amd/scale_replay.py). */
int bj_comm_replay_capture(bj_comm *comm, void *d_capture, size_t capacity_bytes);
int bj_comm_replay_captured(const bj_comm *comm, size_t *bytes);
/* bj_prove / bj_prove_dev on a sharded setup are collective: every rank calls them with the same witness. */
void bj_setup_destroy(bj_setup *s);
int bj_setup_cap(const bj_setup *s, uint64_t *h_cap); /* vk.setup_merkle_tree_cap: cap_size*4 u64 */
/* HBM held by a setup (its shard on a sharded setup): natural-order columns, monomials, the LDE of the cosets it owns, its
 * Merkle subtree, 1 / (x - 1) on its quotient points — the "per rank" column of DESIGN.md §6's memory table. */
int bj_setup_device_bytes(const bj_setup *s, size_t *bytes);
int bj_setup_shape(const bj_setup *s, unsigned *log_n, unsigned *num_vars, unsigned *num_witness_cols,
                   unsigned *num_public_inputs); /* any out pointer may be NULL */

/* ---- seam S1 without linking: the reference's own `MemcopySerializable` dumps (src/cs/implementations/fast_serialization.rs) ----
 * A Rust host writes SetupBaseStorage (polynomial_storage.rs:77-126), WitnessVec (witness.rs:29-71) and DenseVariablesCopyHint
 * (hints/mod.rs:10-61) with `write_into_buffer` and hands the bytes over; byte layouts: era_boojum_amd/csrc/dumps.hip (the
 * reference holds no golden bytes for them: layouts follow the code, parity unpinned by vectors).
 * bj_setup_create_from_dump: `circuit` carries what is code on the Rust side — geometry, the gates in gate_idx order (kind,
 * repetitions, strides, terms, op list), public input locations.  Taken from the dump instead of the struct: the columns,
 * num_constant_cols, table_id_col, every gate's selector path (TreeNode: left = the constant column), quotient_degree when the
 * field is 0 (from the tree's depth + degree), non_residues when the pointer is NULL (make_non_residues, utils.rs:636-688). */
int bj_setup_create_from_dump(bj_ctx *ctx, const bj_circuit *circuit, const void *setup_base, size_t setup_base_len,
                              const bj_proof_config *config, bj_setup **out);
/* What a SetupBaseStorage dump holds, without a device: info8 = {rows n, copy-permutation polynomials, constant columns, table
 * columns, table-id columns, first table-id column, gate indices named by the selector tree (highest + 1), max (depth + degree)
 * in the low word | longest selector path << 32}.  BJ_ERR_INVALID_ARG for bytes that do not parse.  Host-only. */
int bj_setup_dump_info(const void *setup_base, size_t setup_base_len, uint64_t *info8);
/* witness_set_from_witness_vec (witness.rs:386-443) on the device — cell = all_values[hint & (2^48 - 1)], 0 where bit 63 marks a
 * placeholder; multiplicities zero-extended to the trace (witness.rs:225-272); public input values read from their cells — then
 * bj_prove_dev.  witness_hint: the DenseWitnessCopyHint dump (hints/mod.rs:17-21; same layout and indexing) when the circuit has
 * non-copiable witness columns, NULL / 0 otherwise. */
int bj_prove_from_dumps(bj_ctx *ctx, const bj_setup *setup, const void *witness_vec, size_t witness_vec_len,
                        const void *variables_hint, size_t variables_hint_len, const void *witness_hint, size_t witness_hint_len,
                        bj_proof **out);

/* WitnessSet (witness.rs:21-27) in: variables, then the non-copiable witness columns: [num_vars + num_witness_cols][n] natural
 * order, multiplicities [n] (NULL without lookups),
 * public input values in location order.  Proof out (bj_proof_serialize).  Returns BJ_ERR_INVALID_ARG with
 * "constraint system is not satisfied" where the reference panics "unsatisfied" (prover.rs:1425-1438). */
int bj_prove(bj_ctx *ctx, const bj_setup *setup, const uint64_t *h_variables, const uint64_t *h_multiplicities,
             const uint64_t *h_public_values, bj_proof **out);
/* same with the witness already resident in HBM ([num_vars + num_witness_cols][n] contiguous; not modified) */
int bj_prove_dev(bj_ctx *ctx, const bj_setup *setup, const uint64_t *d_variables, const uint64_t *d_multiplicities,
                 const uint64_t *h_public_values, bj_proof **out);
/* The host loop over witnesses around prove_cpu_basic (prover.rs:153-168, convenience.rs:119-196), pipelined from ONE host
 * thread: bj_prove_async queues bj_prove(setup, witness) on one of the context's two internal lanes (each its own HIP stream,
 * workspace and witness staging, driven by a library-owned worker thread; created on first use) and returns at once;
 * bj_proof_wait blocks until that proof is complete, hands it over (bj_proof_destroy as usual) and frees the ticket.  With
 *     t[0] = async(w[0]);  for k = 1..: t[k] = async(w[k]); wait(t[k-1]);
 * the PCIe transfer, inverse transforms and first leaf absorptions of proof k run under the latency-bound tail (node layers,
 * FRI tail, transcript round trips) of proof k-1: throughput of the drop-in call >= the rate of bj_prove_dev on a resident
 * witness; every proof is byte for byte what bj_prove returns.  At most two proofs are in flight — a third submission blocks
 * until the lane it is due on is free.  h_variables / h_multiplicities (pinned host memory for full PCIe speed) must stay
 * valid and unchanged until bj_proof_wait returns; h_public_values is copied.  Errors of the proof (unsatisfied witness, out
 * of memory) come back from bj_proof_wait with the text in bj_last_error(ctx).  Every ticket must be waited for exactly once.
 * Memory: each lane holds a workspace of its own (bj_proof_workspace_bytes); bj_ctx_release_workspace frees the lanes' too.
 * Single-device setups only (a sharded proof is a collective: its ranks are already concurrent). */
typedef struct bj_ticket bj_ticket;
int bj_prove_async(bj_ctx *ctx, const bj_setup *setup, const uint64_t *h_variables, const uint64_t *h_multiplicities,
                   const uint64_t *h_public_values, bj_ticket **out);
int bj_proof_wait(bj_ticket *ticket, bj_proof **out);
int bj_proof_poll(const bj_ticket *ticket); /* 1: complete (bj_proof_wait will not block), 0: still running */
void bj_proof_destroy(bj_proof *p);
/* Flat little-endian u64 serialisation of Proof (proof.rs:121-136); layout documented in era_boojum_amd/proof_format.py:
 * header | schedule | public inputs | witness / stage-2 / quotient caps | values at z, z*omega, 0 | FRI caps |
 * final monomials | per query: index, 4 base-oracle openings (leaf elements + path), FRI openings. */
size_t bj_proof_size_u64(const bj_proof *p);
int bj_proof_serialize(const bj_proof *p, uint64_t *out);
/* wall-clock per stage in ms, named after the reference's log lines: [0] witness LDE + tree, [1] second stage,
 * [2] quotient work and LDE + tree, [3] openings at z, [4] batched FRI opening computation (DEEP), [5] FRI, [6] queries;
 * [7] = duration of the witness-tree Poseidon2 leaf kernel alone, measured with HIP events on the launch stream */
int bj_proof_stage_ms(const bj_proof *p, float *out8);
/* Per-kernel measurements of the proof (measurement only; SURVEY §8d "each evidenced by ... HBM GB/s against the roofline"): the
 * FIRST launch inside this proof of each probed kernel — "quotient_gates" (prover.rs:1031-1080), "quotient_copy_perm"
 * (copy_permutation.rs:1000-1249), "barycentric_eval" (utils.rs:907-1242; the set at z), "deep_accumulate_multi"
 * (prover.rs:2523-2706), "fri_fold_first" (fri/mod.rs:362-678) — bracketed by HIP events on the launch stream: its duration in
 * ms and the algorithmic bytes of that launch by SURVEY §8d's per-unit figures.  index = 0, 1, ... until BJ_ERR_INVALID_ARG;
 * *name points to a string with static storage.  Any out pointer may be NULL. */
int bj_proof_kernel_stats(const bj_proof *p, unsigned index, const char **name, float *ms, double *algorithmic_bytes);
/* Sharded proofs: how long this rank spent inside collectives (sum over the all-gathers of the time between their start and
 * their end on the proof's stream: transfer + waiting for the slowest peer), how many there were and how many bytes arrived
 * from the other ranks.  Zeroes for a single-GPU proof.  Any out pointer may be NULL. */
int bj_proof_comm_stats(const bj_proof *p, float *ms_in_collectives, size_t *calls, size_t *bytes_received);
/* Workspace of the proof: bytes the context had reserved for it (one bump-allocated arena, sized from the setup's geometry
 * before the first kernel), the high-water mark the proof reached, and the number of overflow slabs it had to take because the
 * reservation was too small (0 on every configuration the test suite proves; a slab is a hipMalloc in the middle of a proof, and
 * the context reserves the learned size from the next proof on).  With bj_setup_device_bytes: what a rank holds. */
int bj_proof_workspace_bytes(const bj_proof *p, size_t *reserved, size_t *high_water, size_t *overflow_slabs);

#ifdef __cplusplus
}
#endif
#endif /* BOOJUM_HIP_H */
