// Poseidon2 (Goldilocks, t = 12, rate 8 / capacity 4, x^7, 4 + 22 + 4 rounds) Merkle-tree hashing for gfx950.
//
// Must equal the reference's CPU tree hasher bit for bit (as canonical residues):
//   permutation        src/implementations/poseidon2/state_generic_impl.rs:128-233
//   external matrix    src/implementations/suggested_mds.rs:21-103      circ(2*M4, M4, M4)
//   internal matrix    src/implementations/poseidon2/params.rs:38-39    1 + diag(2^{4,14,11,8,0,5,2,9,13,6,3,12})
//   sponge             src/algebraic_props/sponge.rs:224-346            overwrite absorption, zero-padded tail, no length tag
//   leaf / node hash   src/cs/oracle/mod.rs:114-176
//   tree               src/cs/oracle/merkle_tree.rs:78-174 (construct), 176-386 (chunked), 388-449 (node layers)
//
// Mapping: one lane = one leaf (or one parent node).  The 12-word sponge state lives in VGPRs for the whole leaf;
// for leaf I the lane reads element I of every column, so a wavefront reads 64 consecutive u64 of one column per
// load (512 B, coalesced).  Round constants are wave-uniform and come through the scalar cache.
// The work is integer-ALU bound (~472 field multiplications per 64 absorbed bytes), not HBM bound.
#include "gl.cuh"
#include "kernels.h"
#include "poseidon_rc.inc"

using gl::u64;
using gl::u32;

namespace bj {

__constant__ u64 POSEIDON_RC[BJ_POSEIDON_NUM_RC] = BJ_POSEIDON_RC_TABLE;

__device__ __forceinline__ u64 pow7(u64 x) {
    u64 x2 = gl::sqr(x), x3 = gl::mul(x2, x), x4 = gl::sqr(x2);
    return gl::mul(x4, x3);
}

// M4 block: [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]]
__device__ __forceinline__ void m4(u64 &x0, u64 &x1, u64 &x2, u64 &x3) {
    u64 t0 = gl::add(x0, x1), t1 = gl::add(x2, x3);
    u64 t2 = gl::add(gl::dbl(x1), t1), t3 = gl::add(gl::dbl(x3), t0);
    u64 t4 = gl::add(gl::dbl(gl::dbl(t1)), t3), t5 = gl::add(gl::dbl(gl::dbl(t0)), t2);
    x0 = gl::add(t3, t5);
    x1 = t5;
    x2 = gl::add(t2, t4);
    x3 = t4;
}

__device__ __forceinline__ void ext_mds(u64 (&s)[12]) {
    m4(s[0], s[1], s[2], s[3]);
    m4(s[4], s[5], s[6], s[7]);
    m4(s[8], s[9], s[10], s[11]);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        u64 sum = gl::add(gl::add(s[j], s[4 + j]), s[8 + j]);
        s[j] = gl::add(s[j], sum);
        s[4 + j] = gl::add(s[4 + j], sum);
        s[8 + j] = gl::add(s[8 + j], sum);
    }
}

__device__ __forceinline__ void poseidon2_permutation(u64 (&s)[12]) {
    constexpr unsigned SH[12] = {4, 14, 11, 8, 0, 5, 2, 9, 13, 6, 3, 12};
    ext_mds(s);
    int r = 0;
#pragma unroll 1
    for (int i = 0; i < 4; i++, r++) {
#pragma unroll
        for (int k = 0; k < 12; k++) s[k] = pow7(gl::add(s[k], POSEIDON_RC[12 * r + k]));
        ext_mds(s);
    }
#pragma unroll 1
    for (int i = 0; i < 22; i++, r++) {
        s[0] = pow7(gl::add(s[0], POSEIDON_RC[12 * r]));
        u64 sum = s[0];
#pragma unroll
        for (int k = 1; k < 12; k++) sum = gl::add(sum, s[k]);
#pragma unroll
        for (int k = 0; k < 12; k++) s[k] = gl::add(gl::mul_pow2(s[k], SH[k]), sum);
    }
#pragma unroll 1
    for (int i = 0; i < 4; i++, r++) {
#pragma unroll
        for (int k = 0; k < 12; k++) s[k] = pow7(gl::add(s[k], POSEIDON_RC[12 * r + k]));
        ext_mds(s);
    }
}

// ---------------------------------------------------------------------------------------------------------
// leaf hashing: leaf I = sponge(cols[0][I], cols[1][I], ...)             (merkle_tree.rs:78-174)
// cols given either as base + c*stride or through a device array of column pointers.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
poseidon2_leaves_kernel(const u64 *base, size_t col_stride, const u64 *const *col_ptrs, unsigned n_cols,
                        size_t num_leaves, u64 *digests) {
    size_t I = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= num_leaves) return;
    u64 s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = 0;
    unsigned c = 0;
    for (; c + 8 <= n_cols; c += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const u64 *p = col_ptrs ? col_ptrs[c + k] : base + (size_t)(c + k) * col_stride;
            s[k] = gl::canon(p[I]);
        }
        poseidon2_permutation(s);
    }
    if (c < n_cols) {
        unsigned rem = n_cols - c;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if ((unsigned)k < rem) {
                const u64 *p = col_ptrs ? col_ptrs[c + k] : base + (size_t)(c + k) * col_stride;
                s[k] = gl::canon(p[I]);
            } else {
                s[k] = 0;
            }
        }
        poseidon2_permutation(s);
    }
    // digest = state[0..4]; 32 B per lane
    ulonglong2 *d = reinterpret_cast<ulonglong2 *>(digests + 4 * I);
    d[0] = make_ulonglong2(s[0], s[1]);
    d[1] = make_ulonglong2(s[2], s[3]);
}

// leaf j = sponge( src0[jE..(j+1)E) || src1[jE..(j+1)E) || ... )           (merkle_tree.rs:176-386, FRI oracles)
__global__ void __launch_bounds__(256)
poseidon2_leaves_chunked_kernel(const u64 *src0, const u64 *src1, unsigned n_srcs, unsigned log_e, size_t num_leaves,
                                u64 *digests) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= num_leaves) return;
    const unsigned E = 1u << log_e;
    const unsigned total = n_srcs * E;
    u64 s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = 0;
    unsigned filled = 0;
    for (unsigned t = 0; t < total; t++) {
        unsigned src = t >> log_e, off = t & (E - 1);
        const u64 *p = src == 0 ? src0 : src1;
        u64 v = gl::canon(p[j * E + off]);
        // filled is wave-uniform; write through a switch to keep the state in registers
        switch (filled) {
            case 0: s[0] = v; break;
            case 1: s[1] = v; break;
            case 2: s[2] = v; break;
            case 3: s[3] = v; break;
            case 4: s[4] = v; break;
            case 5: s[5] = v; break;
            case 6: s[6] = v; break;
            default: s[7] = v; break;
        }
        if (++filled == 8) {
            poseidon2_permutation(s);
            filled = 0;
        }
    }
    if (filled) {
        switch (filled) {  // zero-pad the rate part
            case 1: s[1] = 0; [[fallthrough]];
            case 2: s[2] = 0; [[fallthrough]];
            case 3: s[3] = 0; [[fallthrough]];
            case 4: s[4] = 0; [[fallthrough]];
            case 5: s[5] = 0; [[fallthrough]];
            case 6: s[6] = 0; [[fallthrough]];
            default: s[7] = 0;
        }
        poseidon2_permutation(s);
    }
    ulonglong2 *d = reinterpret_cast<ulonglong2 *>(digests + 4 * j);
    d[0] = make_ulonglong2(s[0], s[1]);
    d[1] = make_ulonglong2(s[2], s[3]);
}

// node layer: parent i = perm(left || right || 0000)[0..4]                 (oracle/mod.rs:162-168)
__global__ void __launch_bounds__(256) poseidon2_nodes_kernel(const u64 *children, u64 *parents, size_t num_parents) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_parents) return;
    const ulonglong2 *c = reinterpret_cast<const ulonglong2 *>(children + 8 * i);
    ulonglong2 a = c[0], b = c[1], e = c[2], f = c[3];
    u64 s[12] = {gl::canon(a.x), gl::canon(a.y), gl::canon(b.x), gl::canon(b.y),
                 gl::canon(e.x), gl::canon(e.y), gl::canon(f.x), gl::canon(f.y), 0, 0, 0, 0};
    poseidon2_permutation(s);
    ulonglong2 *d = reinterpret_cast<ulonglong2 *>(parents + 4 * i);
    d[0] = make_ulonglong2(s[0], s[1]);
    d[1] = make_ulonglong2(s[2], s[3]);
}

__global__ void poseidon2_permute_states_kernel(u64 *states, size_t n_states) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_states) return;
    u64 s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = gl::canon(states[12 * i + k]);
    poseidon2_permutation(s);
#pragma unroll
    for (int k = 0; k < 12; k++) states[12 * i + k] = s[k];
}

void launch_poseidon2_leaves(const u64 *d_base, size_t col_stride, const u64 *const *d_col_ptrs, unsigned n_cols,
                             size_t num_leaves, u64 *d_digests, hipStream_t s) {
    unsigned tpb = 256;
    hipLaunchKernelGGL(poseidon2_leaves_kernel, dim3((unsigned)((num_leaves + tpb - 1) / tpb)), dim3(tpb), 0, s,
                       d_base, col_stride, d_col_ptrs, n_cols, num_leaves, d_digests);
}
void launch_poseidon2_leaves_chunked(const u64 *d_src0, const u64 *d_src1, unsigned n_srcs, unsigned log_e,
                                     size_t num_leaves, u64 *d_digests, hipStream_t s) {
    unsigned tpb = 256;
    hipLaunchKernelGGL(poseidon2_leaves_chunked_kernel, dim3((unsigned)((num_leaves + tpb - 1) / tpb)), dim3(tpb), 0,
                       s, d_src0, d_src1, n_srcs, log_e, num_leaves, d_digests);
}
// tree layout: layer 0 = num_leaves digests, then num_leaves/2, ... down to cap_size (inclusive), back to back
void launch_poseidon2_node_layers(u64 *d_tree, size_t num_leaves, size_t cap_size, hipStream_t s) {
    u64 *prev = d_tree;
    size_t len = num_leaves;
    while (len > cap_size) {
        u64 *next = prev + 4 * len;
        size_t nl = len / 2;
        unsigned tpb = 256;
        hipLaunchKernelGGL(poseidon2_nodes_kernel, dim3((unsigned)((nl + tpb - 1) / tpb)), dim3(tpb), 0, s, prev, next,
                           nl);
        prev = next;
        len = nl;
    }
}
void launch_poseidon2_permute_states(u64 *d_states, size_t n_states, hipStream_t s) {
    unsigned tpb = 64;
    hipLaunchKernelGGL(poseidon2_permute_states_kernel, dim3((unsigned)((n_states + tpb - 1) / tpb)), dim3(tpb), 0, s,
                       d_states, n_states);
}

}  // namespace bj
