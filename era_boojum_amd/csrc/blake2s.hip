// Blake2s-256 Merkle hashing for gfx950 — the tree hasher of the reference's non-recursive configuration
// (`impl TreeHasher<F> for blake2::Blake2s256`, src/cs/oracle/mod.rs:179-245; bench `run_sha256_prover_non_recursive`,
// gadgets/sha256/mod.rs:265-270).  The hash itself is RFC 7693 (crate blake2 0.10, unkeyed, 32-byte digest):
//   leaf  = Blake2s( le64(canonical(e_0)) || le64(canonical(e_1)) || ... )        one update per element
//   node  = Blake2s( left[32] || right[32] )
// lane = leaf / node, the 8-word chaining value and the 16-word message block live in VGPRs; a block is 8 field
// elements, i.e. 8 coalesced column loads.  Pure 32-bit add/xor/rotate (v_alignbit) work: ~1.3 k VALU instructions per
// 64-byte block against ~14 k for one Poseidon2 permutation over the same 8 elements.
// Digests are stored as four little-endian u64 words = the 32 digest bytes in memory order (never canonicalised).
#include "gl.h"
#include "kernels.h"
#include "../../include/boojum_hip.h"

using gl::u64;
using gl::u32;

namespace bj {
namespace {

__device__ __constant__ const u32 B2S_IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                                               0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
__device__ __constant__ const unsigned char B2S_SIGMA[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};

__device__ __forceinline__ u32 rotr(u32 x, unsigned n) { return __builtin_amdgcn_alignbit(x, x, n); }

struct B2s {
    u32 h[8];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < 8; i++) h[i] = B2S_IV[i];
        h[0] ^= 0x01010020u;   // digest length 32, no key, fanout 1, depth 1
    }
    // one compression (RFC 7693 §3.2); t = bytes hashed so far including this block
    __device__ __forceinline__ void compress(const u32 (&m)[16], u64 t, bool last) {
        u32 v[16];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            v[i] = h[i];
            v[8 + i] = B2S_IV[i];
        }
        v[12] ^= (u32)t;
        v[13] ^= (u32)(t >> 32);
        if (last) v[14] = ~v[14];
#define B2S_G(a, b, c, d, x, y)                 \
    v[a] = v[a] + v[b] + (x);                   \
    v[d] = rotr(v[d] ^ v[a], 16);               \
    v[c] = v[c] + v[d];                         \
    v[b] = rotr(v[b] ^ v[c], 12);               \
    v[a] = v[a] + v[b] + (y);                   \
    v[d] = rotr(v[d] ^ v[a], 8);                \
    v[c] = v[c] + v[d];                         \
    v[b] = rotr(v[b] ^ v[c], 7);
#pragma unroll
        for (int r = 0; r < 10; r++) {
            const unsigned char *s = B2S_SIGMA[r];
            B2S_G(0, 4, 8, 12, m[s[0]], m[s[1]])
            B2S_G(1, 5, 9, 13, m[s[2]], m[s[3]])
            B2S_G(2, 6, 10, 14, m[s[4]], m[s[5]])
            B2S_G(3, 7, 11, 15, m[s[6]], m[s[7]])
            B2S_G(0, 5, 10, 15, m[s[8]], m[s[9]])
            B2S_G(1, 6, 11, 12, m[s[10]], m[s[11]])
            B2S_G(2, 7, 8, 13, m[s[12]], m[s[13]])
            B2S_G(3, 4, 9, 14, m[s[14]], m[s[15]])
        }
#undef B2S_G
#pragma unroll
        for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[8 + i];
    }
    __device__ __forceinline__ void store(u64 *dst) const {
        ulonglong2 *d = reinterpret_cast<ulonglong2 *>(dst);
        d[0] = make_ulonglong2(gl::pack(h[0], h[1]), gl::pack(h[2], h[3]));
        d[1] = make_ulonglong2(gl::pack(h[4], h[5]), gl::pack(h[6], h[7]));
    }
};

// leaf I = Blake2s over the canonical little-endian bytes of cols[0][I], cols[1][I], ...
__global__ void __launch_bounds__(256)
blake2s_leaves_kernel(const u64 *base, size_t col_stride, const u64 *const *col_ptrs, unsigned n_cols, size_t num_leaves,
                      u64 *digests) {
    const size_t I = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= num_leaves) return;
    B2s st;
    st.init();
    const unsigned n_blocks = n_cols ? (n_cols + 7) / 8 : 1;   // an empty message is one (final) zero block
    for (unsigned b = 0; b < n_blocks; b++) {
        u32 m[16];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const unsigned c = b * 8 + k;
            u64 v = 0;
            if (c < n_cols) {
                const u64 *p = col_ptrs ? col_ptrs[c] : base + (size_t)c * col_stride;
                v = gl::canon(p[I]);
            }
            m[2 * k] = gl::lo32(v);
            m[2 * k + 1] = gl::hi32(v);
        }
        const bool last = b + 1 == n_blocks;
        const u64 t = last ? (u64)n_cols * 8 : (u64)(b + 1) * 64;
        st.compress(m, t, last);
    }
    st.store(digests + 4 * I);
}

// leaf j = Blake2s( src0[jE..(j+1)E) || src1[jE..(j+1)E) )   (FRI oracles, merkle_tree.rs:176-386)
__global__ void __launch_bounds__(256)
blake2s_leaves_chunked_kernel(const u64 *src0, const u64 *src1, unsigned n_srcs, unsigned log_e, size_t num_leaves,
                              u64 *digests) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= num_leaves) return;
    const unsigned E = 1u << log_e, total = n_srcs * E;
    B2s st;
    st.init();
    const unsigned n_blocks = (total + 7) / 8;
    for (unsigned b = 0; b < n_blocks; b++) {
        u32 m[16];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const unsigned e = b * 8 + k;
            u64 v = 0;
            if (e < total) {
                const u64 *p = (e >> log_e) == 0 ? src0 : src1;
                v = gl::canon(p[j * E + (e & (E - 1))]);
            }
            m[2 * k] = gl::lo32(v);
            m[2 * k + 1] = gl::hi32(v);
        }
        const bool last = b + 1 == n_blocks;
        st.compress(m, last ? (u64)total * 8 : (u64)(b + 1) * 64, last);
    }
    st.store(digests + 4 * j);
}

// next[i] = Blake2s( prev[2i] || prev[2i+1] ): exactly one 64-byte block
__global__ void __launch_bounds__(256) blake2s_nodes_kernel(const u64 *prev, u64 *next, size_t n_nodes) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(prev + 8 * i);
    u32 m[16];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        ulonglong2 w = p[k];
        m[4 * k] = gl::lo32(w.x);
        m[4 * k + 1] = gl::hi32(w.x);
        m[4 * k + 2] = gl::lo32(w.y);
        m[4 * k + 3] = gl::hi32(w.y);
    }
    B2s st;
    st.init();
    st.compress(m, 64, true);
    st.store(next + 4 * i);
}

// Proof of work (impl PoWRunner for Blake2s256, src/cs/implementations/pow.rs:50-133): the smallest nonce such that the
// first 8 digest bytes of Blake2s(seed || le64(nonce)), read as a little-endian u64, have >= pow_bits trailing zeros.
// seed = 5 field elements = 40 bytes, so seed || nonce is one 48-byte block.  lane = nonce; the minimum over the launch.
struct PowSeed {
    u32 w[10];
};
__global__ void __launch_bounds__(256) blake2s_pow_kernel(PowSeed seed, unsigned pow_bits, u64 base, u64 count, unsigned long long *result) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const u64 nonce = base + i;
    u32 m[16];
#pragma unroll
    for (int k = 0; k < 10; k++) m[k] = seed.w[k];
    m[10] = gl::lo32(nonce);
    m[11] = gl::hi32(nonce);
    m[12] = m[13] = m[14] = m[15] = 0;
    B2s st;
    st.init();
    st.compress(m, 48, true);
    const u64 first = gl::pack(st.h[0], st.h[1]);
    const unsigned tz = first ? (unsigned)__builtin_ctzll(first) : 64u;
    if (tz >= pow_bits) atomicMin(result, (unsigned long long)nonce);
}

}  // namespace

void launch_blake2s_pow(const u64 *seed5, unsigned pow_bits, u64 base, u64 count, u64 *d_result, hipStream_t s) {
    PowSeed ps;
    for (int k = 0; k < 5; k++) {
        ps.w[2 * k] = gl::lo32(seed5[k]);
        ps.w[2 * k + 1] = gl::hi32(seed5[k]);
    }
    hipLaunchKernelGGL(blake2s_pow_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, ps, pow_bits, base, count,
                       (unsigned long long *)d_result);
}

void launch_blake2s_leaves(const u64 *d_base, size_t col_stride, const u64 *const *d_col_ptrs, unsigned n_cols,
                           size_t num_leaves, u64 *d_digests, hipStream_t s) {
    hipLaunchKernelGGL(blake2s_leaves_kernel, dim3((unsigned)((num_leaves + 255) / 256)), dim3(256), 0, s, d_base, col_stride,
                       d_col_ptrs, n_cols, num_leaves, d_digests);
}
void launch_blake2s_leaves_chunked(const u64 *d_src0, const u64 *d_src1, unsigned n_srcs, unsigned log_e, size_t num_leaves,
                                   u64 *d_digests, hipStream_t s) {
    hipLaunchKernelGGL(blake2s_leaves_chunked_kernel, dim3((unsigned)((num_leaves + 255) / 256)), dim3(256), 0, s, d_src0,
                       d_src1, n_srcs, log_e, num_leaves, d_digests);
}
void launch_blake2s_node_layers(u64 *d_tree, size_t num_leaves, size_t cap_size, hipStream_t s) {
    u64 *prev = d_tree;
    size_t len = num_leaves;
    while (len > cap_size) {
        u64 *next = prev + 4 * len;
        const size_t nl = len / 2;
        hipLaunchKernelGGL(blake2s_nodes_kernel, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, s, prev, next, nl);
        prev = next;
        len = nl;
    }
}

// hasher-dispatching entry points used by the C ABI and the prover (tree layout is the same for both hashers)
void launch_tree_leaves(int hasher, const u64 *d_base, size_t col_stride, const u64 *const *d_col_ptrs, unsigned n_cols,
                        size_t num_leaves, u64 *d_digests, hipStream_t s) {
    if (hasher == BJ_HASHER_BLAKE2S)
        launch_blake2s_leaves(d_base, col_stride, d_col_ptrs, n_cols, num_leaves, d_digests, s);
    else if (hasher == BJ_HASHER_KECCAK256)
        launch_keccak_leaves(d_base, col_stride, d_col_ptrs, n_cols, num_leaves, d_digests, s);
    else
        launch_poseidon2_leaves(d_base, col_stride, d_col_ptrs, n_cols, num_leaves, d_digests, s);
}
void launch_tree_leaves_chunked(int hasher, const u64 *d_src0, const u64 *d_src1, unsigned n_srcs, unsigned log_e,
                                size_t num_leaves, u64 *d_digests, hipStream_t s) {
    if (hasher == BJ_HASHER_BLAKE2S)
        launch_blake2s_leaves_chunked(d_src0, d_src1, n_srcs, log_e, num_leaves, d_digests, s);
    else if (hasher == BJ_HASHER_KECCAK256)
        launch_keccak_leaves_chunked(d_src0, d_src1, n_srcs, log_e, num_leaves, d_digests, s);
    else
        launch_poseidon2_leaves_chunked(d_src0, d_src1, n_srcs, log_e, num_leaves, d_digests, s);
}
void launch_tree_node_layers(int hasher, u64 *d_tree, size_t num_leaves, size_t cap_size, hipStream_t s) {
    if (hasher == BJ_HASHER_BLAKE2S)
        launch_blake2s_node_layers(d_tree, num_leaves, cap_size, s);
    else if (hasher == BJ_HASHER_KECCAK256)
        launch_keccak_node_layers(d_tree, num_leaves, cap_size, s);
    else
        launch_poseidon2_node_layers(d_tree, num_leaves, cap_size, s);
}

}  // namespace bj
