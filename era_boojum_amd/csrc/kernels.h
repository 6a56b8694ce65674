// Internal launch interface between the kernel translation units (*.hip) and the C-ABI layer (boojum_hip.cpp).
// Not part of the public boundary (that is include/boojum_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <string>

namespace bj {
typedef uint64_t u64;

// The environment switches of DESIGN.md §3 (A/B plans and test hooks; none changes a result), read ONCE — by the first
// bj_ctx_create of the process, under std::call_once — and never again on a proof path: no getenv races with a host thread's
// setenv, no per-proof lookups, and every context of a process (one per GPU, each on its own host thread) sees the same plan.
// bj_env_reload() re-reads them for tests that exercise both sides of a switch in one process (not thread-safe by contract).
struct EnvConfig {
    bool ntt_first_narrow = false, ntt_generic = false, ntt_generic_remainder = false, bitrev_gather = false;
    int ntt_front = 4;                  // BJ_NTT_FRONT: 0 remainder passes only, 5 first5 wherever it applies, default 4
    int ntt_first4_v = 2;               // BJ_NTT_FIRST4_V: indices per lane of the four-round front pass
    int ntt_first4_mode = 0;            // BJ_NTT_FIRST4_MODE: 0 inputs of the front pass kept in registers across the cosets (round 3: two waves per SIMD); 3 / 4: re-read per coset (L2), that many waves
    bool ntt_two_pass = true;           // BJ_NTT_TWO_PASS=0: 2^22-point transforms as 4 + 8 + 10 rounds (rounds 3-5) instead of 10 + 12
    bool mono_tiled = true;             // BJ_MONO_TILED=0: bj_prove keeps 2^22-row monomials in natural order (inverse transforms end in a bit-reversal pass)
    bool gate_no_aot = false, gate_no_fuse = false, gate_no_jit = false;
    bool gates_windowed = true;         // BJ_GATES_WINDOWED=0: per-gate kernel for the hand-written kinds
    bool prove_no_absorb = false;
    bool copy_perm_wide_k = false;       // BJ_COPY_PERM_WIDE_K: quotient_copy_perm with 64-bit non-residue products even when they fit 32 bits (A/B, tests)
    bool prove_uniform_groups = false;   // BJ_PROVE_UNIFORM_GROUPS: equal groups of G columns, one multi-block absorption run per group (round 4's GROUPING only: its launches absorbed eight columns each)
    bool async_stagger = true;           // BJ_ASYNC_STAGGER=0: bj_prove_async lanes start whenever they are given work (A/B)
    int async_mode = -1;                 // BJ_ASYNC_MODE: what a bj_prove_async lane does while its sibling proves: -1 by witness size (default), 0 bj_prove as is (+ stagger), 1 whole witness first, 2 groups + one hash
    unsigned prove_h2d_group = 8;
    size_t nodes_lanepar_max = 16384;
    std::string jit_cache_dir, rccl_lib;
};
const EnvConfig &env();
void env_reload();


// ntt.hip
void launch_twiddles(u64 *d_out, unsigned log_n, bool inverse, hipStream_t s);
void launch_round_scales(u64 *d_out, const u64 *h_shifts, unsigned n_cosets, unsigned log_n, hipStream_t s);
// d_front_table: BJ_FRONT_TABLE_WORDS words of device scratch for the twiddle table of the two-pass plan (nullptr: never that plan)
constexpr size_t BJ_FRONT_TABLE_WORDS = 64 * 1024;
void launch_ntt_passes(const u64 *d_in, u64 *d_out, const u64 *d_tw, const u64 *d_round_scale, unsigned log_n,
                       unsigned n_cols, unsigned n_cosets, size_t in_col_stride, size_t out_col_stride, hipStream_t s,
                       u64 *d_front_table = nullptr, bool tiled_in = false);
// true when launch_ntt_passes would take the two-pass plan for these arguments (the only plan that reads the tiled layout)
bool ntt_two_pass_applies(const u64 *d_in, const u64 *d_out, unsigned log_n, unsigned n_cosets, size_t in_col_stride, size_t out_col_stride);
void launch_bitrev_scale(const u64 *d_in, u64 *d_out, unsigned log_n, unsigned n_cols, size_t in_col_stride,
                         size_t out_col_stride, u64 scale, u64 step, hipStream_t s);
void launch_canonicalize(u64 *d, size_t n, hipStream_t s);
void launch_field_op(int op, const u64 *a, const u64 *b, u64 *out, size_t n, hipStream_t s);

// ntt_r16.hip (register-radix-16 passes)
void launch_ntt_local12(const u64 *in, u64 *out, const u64 *tw, const u64 *round_scale, unsigned log_n,
                        unsigned n_cols, unsigned n_cosets, size_t in_col_stride, size_t in_coset_stride,
                        size_t out_col_stride, unsigned rounds /* 12, or 10 / 9 behind launch_ntt_first4 / first5 */, hipStream_t s);
void launch_ntt_front10(const u64 *in, u64 *out, const u64 *tw, const u64 *round_scale, u64 *d_table, unsigned log_n, unsigned n_cols,
                        unsigned n_cosets, size_t in_col_stride, size_t out_col_stride, hipStream_t s, bool tiled_in = false);
// 2^22-word columns in the tiled layout (ntt_r16.hip: tiled_index): last pass of an inverse transform storing it, re-layout kernel
void launch_ntt_local12_pair_tiled(const u64 *in, u64 *out, const u64 *tw, const u64 *tw_scaled /* tw[j] * scale, j < 2^21 */, u64 scale,
                                   unsigned n_cols, size_t in_col_stride, size_t out_col_stride, hipStream_t s);
void launch_scale_table(const u64 *in, u64 *out, size_t count, u64 scale, hipStream_t s);
void launch_tiled_permute(const u64 *in, u64 *out, unsigned n_cols, size_t in_col_stride, size_t out_col_stride, bool to_tiled, hipStream_t s);
void launch_ntt_first5(const u64 *in, u64 *out, const u64 *tw, const u64 *round_scale, unsigned log_n, unsigned n_cols,
                       unsigned n_cosets, size_t in_col_stride, size_t in_coset_stride, size_t out_col_stride, hipStream_t s);
void launch_ntt_first4(const u64 *in, u64 *out, const u64 *tw, const u64 *round_scale, unsigned log_n, unsigned n_cols,
                       unsigned n_cosets, size_t in_col_stride, size_t in_coset_stride, size_t out_col_stride, hipStream_t s);
void launch_ntt_strided8(const u64 *in, u64 *out, const u64 *tw, const u64 *round_scale, unsigned log_n, unsigned r0,
                         unsigned n_cols, unsigned n_cosets, size_t in_col_stride, size_t in_coset_stride,
                         size_t out_col_stride, hipStream_t s);
void launch_ntt_strided4(const u64 *in, u64 *out, const u64 *tw, const u64 *round_scale, unsigned log_n, unsigned r0,
                         unsigned n_cols, unsigned n_cosets, size_t in_col_stride, size_t in_coset_stride,
                         size_t out_col_stride, hipStream_t s);

// poseidon2.hip
void launch_poseidon2_leaves(const u64 *d_base, size_t col_stride, const u64 *const *d_col_ptrs, unsigned n_cols,
                             size_t num_leaves, u64 *d_digests, hipStream_t s);
// one absorption of up to eight columns per leaf; d_capacity [4][num_leaves] carries the sponge between the groups
void launch_poseidon2_leaves_absorb(const u64 *d_base, size_t col_stride, unsigned n_cols, size_t num_leaves, u64 *d_capacity,
                                    u64 *d_digests, bool first, bool last, hipStream_t s);
void launch_poseidon2_leaves_chunked(const u64 *d_src0, const u64 *d_src1, unsigned n_srcs, unsigned log_e,
                                     size_t num_leaves, u64 *d_digests, hipStream_t s);
void launch_poseidon2_node_layers(u64 *d_tree, size_t num_leaves, size_t cap_size, hipStream_t s);
void launch_poseidon2_permute_states(u64 *d_states, size_t n_states, hipStream_t s);
// blake2s.hip: the same tree with Blake2s-256 digests, and the hasher-dispatching entry points (hasher = BJ_HASHER_*)
void launch_tree_leaves(int hasher, const u64 *d_base, size_t col_stride, const u64 *const *d_col_ptrs, unsigned n_cols,
                        size_t num_leaves, u64 *d_digests, hipStream_t s);
void launch_tree_leaves_chunked(int hasher, const u64 *d_src0, const u64 *d_src1, unsigned n_srcs, unsigned log_e,
                                size_t num_leaves, u64 *d_digests, hipStream_t s);
void launch_tree_node_layers(int hasher, u64 *d_tree, size_t num_leaves, size_t cap_size, hipStream_t s);
// keccak.hip
void launch_keccak_leaves(const u64 *d_base, size_t col_stride, const u64 *const *d_col_ptrs, unsigned n_cols,
                          size_t num_leaves, u64 *d_digests, hipStream_t s);
void launch_keccak_leaves_chunked(const u64 *d_src0, const u64 *d_src1, unsigned n_srcs, unsigned log_e, size_t num_leaves,
                                  u64 *d_digests, hipStream_t s);
void launch_keccak_node_layers(u64 *d_tree, size_t num_leaves, size_t cap_size, hipStream_t s);
void launch_keccak_pow(const u64 *seed5, unsigned pow_bits, u64 base, u64 count, u64 *d_result, hipStream_t s);
// Blake2s proof of work over nonces [base, base + count): atomicMin of the valid ones into *d_result (pre-set to ~0)
void launch_blake2s_pow(const u64 *seed5, unsigned pow_bits, u64 base, u64 count, u64 *d_result, hipStream_t s);

// fri.hip
void launch_fri_fold(const u64 *d_c0, const u64 *d_c1, size_t len, u64 *d_o0, u64 *d_o1, const u64 *d_roots,
                     u64 coset_inv, u64 ch0, u64 ch1, hipStream_t s);
void launch_fri_fold_step(const u64 *d_c0, const u64 *d_c1, size_t len, unsigned k, u64 *d_o0, u64 *d_o1,
                          const u64 *d_roots, u64 coset_inv, u64 ch0, u64 ch1, hipStream_t s, size_t j0 = 0);
// openings.hip: several DEEP opening sets in one launch (device-side argument pointers, canonical scalars)
constexpr int DEEP_MAX_SETS = 3;
struct DeepSetHostArgs {
    const u64 *const *d_cols;
    const u64 *d_coefs;     // [n_cols][2]
    unsigned n_cols;
    u64 c0, c1, at0, at1;
};
void launch_deep_accumulate_multi(const DeepSetHostArgs *sets, unsigned n_sets, size_t N, size_t I0, const u64 *d_tw_fwd,
                                  u64 *d_dst0, u64 *d_dst1, int accumulate, hipStream_t s);
}  // namespace bj
