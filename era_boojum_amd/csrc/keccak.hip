// Keccak-256 Merkle hashing for gfx950 — `impl TreeHasher<F> for sha3::Keccak256` (src/cs/oracle/mod.rs:247-312): the
// original Keccak padding (0x01 ... 0x80, NOT the 0x06 of FIPS-202 SHA3-256), rate 136 bytes, 32-byte digest.
//   leaf = Keccak256( le64(canonical(e_0)) || le64(canonical(e_1)) || ... ),   node = Keccak256( left[32] || right[32] )
// lane = leaf / node; the 25-lane state lives in VGPRs (50 registers); a rate block is 17 field elements = 17 state lanes,
// so absorbing is 17 coalesced column loads XORed straight into the state.  Digests: state lanes 0..3 (little endian).
#include "gl.h"
#include "kernels.h"
#include "../../include/boojum_hip.h"

using gl::u64;
using gl::u32;

namespace bj {
namespace {

__device__ __constant__ const u64 KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

template <unsigned N>
__device__ __forceinline__ u64 rotl64(u64 x) {
    if (N == 0) return x;
    return (x << (N & 63)) | (x >> ((64 - N) & 63));
}

// Keccak-f[1600], lanes a[x + 5y]
__device__ __forceinline__ void keccak_f(u64 (&a)[25]) {
#pragma unroll 1
    for (int round = 0; round < 24; round++) {
        u64 c[5], d[5];
#pragma unroll
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rotl64<1>(c[(x + 1) % 5]);
#pragma unroll
        for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
        // rho + pi: b[y + 5*((2x + 3y) % 5)] = rotl(a[x + 5y], r[x][y])
        u64 b[25];
        b[0] = a[0];
        b[10] = rotl64<1>(a[1]);   b[20] = rotl64<62>(a[2]);  b[5] = rotl64<28>(a[3]);   b[15] = rotl64<27>(a[4]);
        b[16] = rotl64<36>(a[5]);  b[1] = rotl64<44>(a[6]);   b[11] = rotl64<6>(a[7]);   b[21] = rotl64<55>(a[8]);
        b[6] = rotl64<20>(a[9]);   b[7] = rotl64<3>(a[10]);   b[17] = rotl64<10>(a[11]); b[2] = rotl64<43>(a[12]);
        b[12] = rotl64<25>(a[13]); b[22] = rotl64<39>(a[14]); b[23] = rotl64<41>(a[15]); b[8] = rotl64<45>(a[16]);
        b[18] = rotl64<15>(a[17]); b[3] = rotl64<21>(a[18]);  b[13] = rotl64<8>(a[19]);  b[14] = rotl64<18>(a[20]);
        b[24] = rotl64<2>(a[21]);  b[9] = rotl64<61>(a[22]);  b[19] = rotl64<56>(a[23]); b[4] = rotl64<14>(a[24]);
#pragma unroll
        for (int y = 0; y < 25; y += 5)
#pragma unroll
            for (int x = 0; x < 5; x++) a[y + x] = b[y + x] ^ (~b[y + (x + 1) % 5] & b[y + (x + 2) % 5]);
        a[0] ^= KECCAK_RC[round];
    }
}

struct Keccak {
    u64 a[25];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < 25; i++) a[i] = 0;
    }
    // pad10*1 of the original Keccak for a message of `lanes_in_last` whole 8-byte words in the final rate block
    __device__ __forceinline__ void pad_and_permute(unsigned lanes_in_last) {
        u64 first = 0x01ULL, last = 0x8000000000000000ULL;
#pragma unroll
        for (int i = 0; i < 17; i++)
            if ((unsigned)i == lanes_in_last) a[i] ^= first;
        a[16] ^= last;
        keccak_f(a);
    }
    __device__ __forceinline__ void store(u64 *dst) const {
        ulonglong2 *d = reinterpret_cast<ulonglong2 *>(dst);
        d[0] = make_ulonglong2(a[0], a[1]);
        d[1] = make_ulonglong2(a[2], a[3]);
    }
};

__global__ void __launch_bounds__(256)
keccak_leaves_kernel(const u64 *base, size_t col_stride, const u64 *const *col_ptrs, unsigned n_cols, size_t num_leaves,
                     u64 *digests) {
    const size_t I = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= num_leaves) return;
    Keccak st;
    st.init();
    unsigned c = 0;
    for (; c + 17 <= n_cols; c += 17) {   // full rate blocks
#pragma unroll
        for (int k = 0; k < 17; k++) {
            const u64 *p = col_ptrs ? col_ptrs[c + k] : base + (size_t)(c + k) * col_stride;
            st.a[k] ^= gl::canon(p[I]);
        }
        keccak_f(st.a);
    }
    const unsigned rem = n_cols - c;     // 0..16 words in the last block, then the padding
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if ((unsigned)k < rem) {
            const u64 *p = col_ptrs ? col_ptrs[c + k] : base + (size_t)(c + k) * col_stride;
            st.a[k] ^= gl::canon(p[I]);
        }
    }
    st.pad_and_permute(rem);
    st.store(digests + 4 * I);
}

__global__ void __launch_bounds__(256)
keccak_leaves_chunked_kernel(const u64 *src0, const u64 *src1, unsigned n_srcs, unsigned log_e, size_t num_leaves,
                             u64 *digests) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= num_leaves) return;
    const unsigned E = 1u << log_e, total = n_srcs * E;
    Keccak st;
    st.init();
    unsigned e0 = 0;
    for (; e0 + 17 <= total; e0 += 17) {
#pragma unroll
        for (int k = 0; k < 17; k++) {
            const unsigned e = e0 + k;
            const u64 *p = (e >> log_e) == 0 ? src0 : src1;
            st.a[k] ^= gl::canon(p[j * E + (e & (E - 1))]);
        }
        keccak_f(st.a);
    }
    const unsigned rem = total - e0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if ((unsigned)k < rem) {
            const unsigned e = e0 + k;
            const u64 *p = (e >> log_e) == 0 ? src0 : src1;
            st.a[k] ^= gl::canon(p[j * E + (e & (E - 1))]);
        }
    }
    st.pad_and_permute(rem);
    st.store(digests + 4 * j);
}

__global__ void __launch_bounds__(256) keccak_nodes_kernel(const u64 *prev, u64 *next, size_t n_nodes) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(prev + 8 * i);
    Keccak st;
    st.init();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        ulonglong2 w = p[k];
        st.a[2 * k] = w.x;
        st.a[2 * k + 1] = w.y;
    }
    st.pad_and_permute(8);   // 64 message bytes = 8 lanes
    st.store(next + 4 * i);
}

// Proof of work (impl PoWRunner for Keccak256, src/cs/implementations/pow.rs:139-230): the smallest nonce such that the first
// 8 digest bytes of Keccak256(seed || le64(nonce)), read as a little-endian u64, have >= pow_bits trailing zeros.  seed = 5 field
// elements = 40 bytes, so seed || nonce is six lanes of one rate block; lane = nonce, the minimum over the launch.
struct KeccakPowSeed {
    u64 w[5];
};
__global__ void __launch_bounds__(256) keccak_pow_kernel(KeccakPowSeed seed, unsigned pow_bits, u64 base, u64 count, unsigned long long *result) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const u64 nonce = base + i;
    Keccak st;
    st.init();
#pragma unroll
    for (int k = 0; k < 5; k++) st.a[k] = seed.w[k];
    st.a[5] = nonce;
    st.pad_and_permute(6);   // 48 message bytes = 6 lanes
    const u64 first = st.a[0];
    const unsigned tz = first ? (unsigned)__builtin_ctzll(first) : 64u;
    if (tz >= pow_bits) atomicMin(result, (unsigned long long)nonce);
}

}  // namespace

void launch_keccak_pow(const u64 *seed5, unsigned pow_bits, u64 base, u64 count, u64 *d_result, hipStream_t s) {
    KeccakPowSeed ps;
    for (int k = 0; k < 5; k++) ps.w[k] = seed5[k];
    hipLaunchKernelGGL(keccak_pow_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, ps, pow_bits, base, count,
                       (unsigned long long *)d_result);
}

void launch_keccak_leaves(const u64 *d_base, size_t col_stride, const u64 *const *d_col_ptrs, unsigned n_cols,
                          size_t num_leaves, u64 *d_digests, hipStream_t s) {
    hipLaunchKernelGGL(keccak_leaves_kernel, dim3((unsigned)((num_leaves + 255) / 256)), dim3(256), 0, s, d_base, col_stride,
                       d_col_ptrs, n_cols, num_leaves, d_digests);
}
void launch_keccak_leaves_chunked(const u64 *d_src0, const u64 *d_src1, unsigned n_srcs, unsigned log_e, size_t num_leaves,
                                  u64 *d_digests, hipStream_t s) {
    hipLaunchKernelGGL(keccak_leaves_chunked_kernel, dim3((unsigned)((num_leaves + 255) / 256)), dim3(256), 0, s, d_src0,
                       d_src1, n_srcs, log_e, num_leaves, d_digests);
}
void launch_keccak_node_layers(u64 *d_tree, size_t num_leaves, size_t cap_size, hipStream_t s) {
    u64 *prev = d_tree;
    size_t len = num_leaves;
    while (len > cap_size) {
        u64 *next = prev + 4 * len;
        const size_t nl = len / 2;
        hipLaunchKernelGGL(keccak_nodes_kernel, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, s, prev, next, nl);
        prev = next;
        len = nl;
    }
}

}  // namespace bj
