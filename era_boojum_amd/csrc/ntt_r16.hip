// Register-radix-16 NTT pass kernels for gfx950 — the fast path of launch_ntt_passes (ntt.hip holds the generic
// fallback and the documentation of the pass decomposition).
//
// Every thread keeps 16 field elements in VGPRs and performs 4 butterfly rounds ("one radix-16 step") on them
// without touching memory; steps are separated by a transpose through LDS.  Three kernels:
//   ntt_local12_kernel   last 12 rounds on a contiguous 4096-element chunk   (3 steps, 3 LDS transposes, the last
//                        one only to make the HBM store coalesced)
//   ntt_strided8_kernel  8 rounds at stride 2^rem_log, tile = 256 x 16 elems (2 steps, 1 LDS transpose; every HBM
//                        access is a 128-byte run)
//   ntt_strided4_kernel  4 rounds at stride 2^rem_log, tile = 16 x 256 elems (1 step, no LDS)
// Twiddles: round r of group k needs T[k] * shift^(n/2^(r+1)).  Per step a thread needs 15 of them
// (1 + 2 + 4 + 8); they are fetched (and coset-scaled) ONCE per workgroup and reused for every column the
// workgroup loops over: the block-uniform ones are staged in LDS, the per-thread ones stay in VGPRs.
// All LDS indices go through pad(l) = l + l/16, which makes every transpose below bank-conflict free for
// ds_read_b64 / ds_write_b64 (64 x 4-byte banks).
#include "gl.h"
#include "kernels.h"

#include <cstdlib>

using gl::u64;
using gl::u32;

namespace bj {

struct R16Args {
    const u64 *in;
    u64 *out;
    const u64 *tw;           // bit-reversed twiddle table
    const u64 *round_scale;  // [n_cosets][32] or nullptr
    unsigned log_n, r0;      // r0 = first round of this pass
    unsigned n_cols, cols_per_block;
    size_t in_col_stride, in_coset_stride, out_col_stride;
};

#ifndef BJ_R16_WAVES
#define BJ_R16_WAVES 3   // min waves per SIMD requested from the register allocator (tuned on MI355X: 2 -> 8.6 ms, 3 -> 7.9 ms, 4 -> 10.9 ms per 93 x 2^20 x 8 LDE)
#endif
static constexpr u32 TILE = 4096;
static constexpr u32 LDS_ELEMS = TILE + TILE / 16;

// A/B builds for the question "what is the parked time of these passes waiting for" (tools/ntt_wait_ab.sh; results are WRONG
// transforms, only their duration means something; never defined in the product build):
//   -DBJ_R16_AB_NOLOAD     the 16 input words of a lane come from a register expression instead of HBM
//   -DBJ_R16_AB_NOSTORE    the HBM stores sit behind a wave-uniform condition that is never true
//   -DBJ_R16_AB_NOBARRIER  no workgroup barrier around the LDS transposes
#ifdef BJ_R16_AB_NOLOAD
#define BJ_R16_LD(base, off, salt) ((u64)(off) * 0x9E3779B97F4A7C15ull + (u64)(salt))
#else
#define BJ_R16_LD(base, off, salt) ld_off(base, off)
#endif
#ifdef BJ_R16_AB_NOSTORE
#define BJ_R16_ST_GUARD(a) if ((a).n_cols == 0xFFFFFFFFu)
#else
#define BJ_R16_ST_GUARD(a)
#endif
// -DBJ_R16_AB_S8_NOMEM: only the strided pass loses its loads and its stores — its butterflies stay: what an LDE would cost if the
// middle pass's 16 n words per column of HBM traffic went away with the arithmetic unchanged (the bound on fusing it into a neighbour)
#ifdef BJ_R16_AB_S8_NOMEM
#define BJ_R16_LD_S8(base, off, salt) ((u64)(off) * 0x9E3779B97F4A7C15ull + (u64)(salt))
#define BJ_R16_ST_GUARD_S8(a) if ((a).n_cols == 0xFFFFFFFFu)
#else
#define BJ_R16_LD_S8(base, off, salt) BJ_R16_LD(base, off, salt)
#define BJ_R16_ST_GUARD_S8(a) BJ_R16_ST_GUARD(a)
#endif
// Software pipeline over the columns of a workgroup (round 5, build switches): the loads of column c + 1 are issued before the stores
// of column c and the barriers stop draining vmcnt (the ISA then waits with vmcnt(22..16) at the top of a column: the sixteen stores
// stay in flight).  Measured on MI355X (profiles/r05_ntt_wait_ab.txt): parked wave-cycles of ntt_local12 24.4 % -> 20.7 %, issue-stall
// cycles up by the same amount, duration 1.410 -> 1.415 ms at equal clocks (1.362 ms in the one run where the part held 1.89 GHz);
// ntt_strided8 0.924 -> 0.959 ms (sixteen more register pairs: 125 -> 157 VGPRs, four resident waves per SIMD become three).  The
// passes are not short of latency cover: they run at the power limit (1.77-1.80 GHz against 2.3 GHz for the hash kernels; without
// their loads or their stores the SAME instruction stream runs at 1.93 / 2.18 GHz and 7-8 % fewer cycles).  Both default to off.
#ifndef BJ_R16_PREFETCH_LOCAL
#define BJ_R16_PREFETCH_LOCAL 0
#endif
#ifndef BJ_R16_PREFETCH_STRIDED
#define BJ_R16_PREFETCH_STRIDED 0
#endif
#ifndef BJ_R16_WIDE_ST
#define BJ_R16_WIDE_ST 0   // 1: ntt_local12 stores 16 bytes per lane (two adjacent words out of LDS) — half the store instructions
#endif
// Workgroup barrier that orders LDS traffic only (for the pipelined pass).  __syncthreads() is a workgroup-scope fence: it drains
// vmcnt too, i.e. waits for every global store of the wave to be acknowledged and for the loads that were issued ahead for the NEXT
// column — the two things the pipeline keeps in flight across the transposes.  All data exchanged between the lanes of a workgroup
// here goes through LDS, so this wave's LDS operations being complete (lgkmcnt) before the barrier is all the ordering they need.
#ifdef BJ_R16_AB_NOBARRIER
#define BJ_R16_SYNC_FULL()
#define BJ_R16_SYNC_LDS()
#else
#define BJ_R16_SYNC_FULL() __syncthreads()
#define BJ_R16_SYNC_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

#if BJ_R16_PREFETCH_LOCAL
#define BJ_R16_SYNC_L() BJ_R16_SYNC_LDS()
#else
#define BJ_R16_SYNC_L() BJ_R16_SYNC_FULL()
#endif
#if BJ_R16_PREFETCH_STRIDED
#define BJ_R16_SYNC_S() BJ_R16_SYNC_LDS()
#else
#define BJ_R16_SYNC_S() BJ_R16_SYNC_FULL()
#endif

__device__ __forceinline__ u32 pad(u32 l) { return l + (l >> 4); }
__device__ __forceinline__ void set_prio(int p) {   // s_setprio takes an immediate: p is a constant after unrolling
    if (p >= 3) __builtin_amdgcn_s_setprio(3);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}
// wave-uniform base pointer + 32-bit per-lane BYTE offset: the shape the compiler turns into `global_load v, v_off, s[base]`
// (the empty asm pins the base in an SGPR pair: without it the optimiser re-associates base + offset into sixteen hoisted
// 64-bit per-lane addresses)
// A wave-uniform GLOBAL pointer the optimiser has lost track of (computed under a branch it treats as divergent): back into an
// SGPR pair, as an address-space-1 pointer — rebuilt from integers a generic pointer would turn the accesses into flat_load /
// flat_store, which count on lgkmcnt as well and would make every LDS wait a wait for HBM.
typedef const u64 __attribute__((address_space(1))) *gcptr;
typedef u64 __attribute__((address_space(1))) *gptr;
__device__ __forceinline__ gcptr uniform_gptr(const u64 *p) {
    const u64 v = reinterpret_cast<u64>(p);
    return (gcptr)gl::pack(__builtin_amdgcn_readfirstlane(gl::lo32(v)), __builtin_amdgcn_readfirstlane(gl::hi32(v)));
}
__device__ __forceinline__ u64 ld_off(gcptr base, u32 byte_off) {
    asm volatile("" : "+s"(base));
    return *(gcptr)((const char __attribute__((address_space(1))) *)base + byte_off);
}
__device__ __forceinline__ void st_off(gptr base, u32 byte_off, u64 v) {
    asm volatile("" : "+s"(base));
    *(gptr)((char __attribute__((address_space(1))) *)base + byte_off) = v;
}
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
typedef u64x2 __attribute__((address_space(1))) *gptr2;
__device__ __forceinline__ void st_off2(gptr base, u32 byte_off, u64 v0, u64 v1) {   // byte_off a multiple of 16
    asm volatile("" : "+s"(base));
    u64x2 v;
    v.x = v0;
    v.y = v1;
    *(gptr2)((char __attribute__((address_space(1))) *)base + byte_off) = v;
}
__device__ __forceinline__ u64 ld_off(const u64 *base, u32 byte_off) {
    asm volatile("" : "+s"(base));
    return *reinterpret_cast<const u64 *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ void st_off(u64 *base, u32 byte_off, u64 v) {
    asm volatile("" : "+s"(base));
    *reinterpret_cast<u64 *>(reinterpret_cast<char *>(base) + byte_off) = v;
}

// tw[(1<<s)-1+g] = T[(kb<<s)+g] * sc[r+s],  s = 0..3, g < 2^s
template <bool SCALED>
__device__ __forceinline__ void load_step_twiddles(u64 (&tw)[15], const u64 *__restrict__ T, u32 kb, unsigned r,
                                                   const u64 *__restrict__ sc) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
        u64 scale = SCALED ? sc[r + s] : 1;
#pragma unroll
        for (int g = 0; g < (1 << s); g++) {
            u64 t = T[((size_t)kb << s) + g];
            if (SCALED) t = gl::mul(t, scale);
            tw[(1 << s) - 1 + g] = t;
        }
    }
}

// 4 rounds on 16 register-resident elements; round s pairs x[i], x[i + (8>>s)] inside groups of 16>>s.  The elements are WEAK
// residues (any u64) from the load of the first pass to the store of the last one: butterflies run two at a time through
// gl::butterfly2_weak, nothing is canonicalised in between (the last pass does it once per element when it stores).
// tw: the step's 15 twiddles, in registers (array) or in LDS (pointer) — indexed by constants after unrolling either way.
// FIRST_S = 2 runs only the last two of the four rounds (the local pass after a front pass that has already done the other two).
// PRIO: the wave lowers its issue priority round by round (3, 2, 1, 0): the SIMD's arbiter serves the oldest wave first, so the waves
// of a workgroup would reach the barrier behind the step one after the other and the last one would compute alone; with the
// priority falling as a wave advances, the wave that is behind is served first and they arrive together (ntt_front10)
template <bool UNIT_FIRST, int FIRST_S = 0, bool PRIO = false>
__device__ __forceinline__ void radix16(u64 (&x)[16], const u64 *tw) {
#pragma unroll
    for (int s = FIRST_S; s < 3; s++) {
        if (PRIO) set_prio(3 - s);
        const int half = 8 >> s;
#pragma unroll
        for (int g = 0; g < (1 << s); g++) {
            const u64 w = (UNIT_FIRST && s == 0) ? 1 : tw[(1 << s) - 1 + g];
#pragma unroll
            for (int j = 0; j < half; j += 2) {
                const int iu = g * 2 * half + j, iv = iu + half;
                if (UNIT_FIRST && s == 0)
                    gl::addsub2_weak(x[iu], x[iv], x[iu + 1], x[iv + 1]);
                else
                    gl::butterfly2_weak(x[iu], x[iv], w, x[iu + 1], x[iv + 1], w);
            }
        }
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int g = 0; g < 8; g += 2)   // last round: neighbours, one twiddle per pair
        gl::butterfly2_weak(x[2 * g], x[2 * g + 1], tw[7 + g], x[2 * g + 2], x[2 * g + 3], tw[8 + g]);
}
template <bool UNIT_FIRST>
__device__ __forceinline__ void radix16_lds(u64 (&x)[16], const u64 *tw) { radix16<UNIT_FIRST>(x, tw); }
// the same four rounds with every output multiplied by s: the last round runs (s u + (s w) v, s u - (s w) v) — the caller passes its
// eight twiddles tw[7..14] already multiplied by s — so the factor costs one product per butterfly, half a product per element
__device__ __forceinline__ void radix16_scaled(u64 (&x)[16], const u64 *tw, u64 s) {
#pragma unroll
    for (int st = 0; st < 3; st++) {
        const int half = 8 >> st;
#pragma unroll
        for (int g = 0; g < (1 << st); g++) {
            const u64 w = tw[(1 << st) - 1 + g];
#pragma unroll
            for (int j = 0; j < half; j += 2) {
                const int iu = g * 2 * half + j, iv = iu + half;
                gl::butterfly2_weak(x[iu], x[iv], w, x[iu + 1], x[iv + 1], w);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < 8; g += 2) {
        x[2 * g] = gl::mul_weak(x[2 * g], s);
        x[2 * g + 2] = gl::mul_weak(x[2 * g + 2], s);
        gl::butterfly2_weak(x[2 * g], x[2 * g + 1], tw[7 + g], x[2 * g + 2], x[2 * g + 3], tw[8 + g]);
    }
}

// block-uniform step twiddles: 15 lanes compute them once, everyone reads them back as LDS broadcasts
template <bool SCALED>
__device__ __forceinline__ void stage_uniform_twiddles(u64 *lds_tw, const u64 *__restrict__ T, u32 kb, unsigned r,
                                                       const u64 *__restrict__ sc) {
    const u32 t = threadIdx.x;
    if (t < 15) {
        const int s = 31 - __clz(t + 1);          // 0,1,1,2,2,2,2,3...
        const int g = (int)(t + 1) - (1 << s);
        u64 v = T[((size_t)kb << s) + g];
        if (SCALED) v = gl::mul(v, sc[r + s]);
        lds_tw[t] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------
// ROUNDS = 12, or 10 / 9: the same kernel without the first two / three rounds of step A (a 4096-element chunk is then four /
// eight independent sub-transforms; the rounds that remain are exactly the last ROUNDS rounds, with the twiddle indices unchanged)
template <bool SCALED, int ROUNDS = 12>
__global__ void __launch_bounds__(256, BJ_R16_WAVES) ntt_local12_kernel(R16Args a) {
    __shared__ u64 lds[LDS_ELEMS + 16 + 16 * 16];
    u64 *lds_tw = lds + LDS_ELEMS;
    u64 *lds_twB = lds_tw + 16;               // [t >> 4][15 (+1 pad)] twiddles of step B (shared by 16 lanes each)
    const u32 t = threadIdx.x;
    const u32 b = blockIdx.x;
    const unsigned coset = blockIdx.z;
    const unsigned r0 = a.log_n - 12;
    const size_t n = (size_t)1 << a.log_n;
    const u64 *sc = SCALED ? a.round_scale + (size_t)coset * 32 : nullptr;

    u64 twC[15];
    load_step_twiddles<SCALED>(twC, a.tw, b * 256 + t, r0 + 8, sc);
    stage_uniform_twiddles<SCALED>(lds_tw, a.tw, b, r0, sc);
    if (t < 240) {
        const u32 m = t / 15, i = t % 15;
        const int s = 31 - __clz(i + 1);
        const int g = (int)(i + 1) - (1 << s);
        u64 v = a.tw[(((size_t)b * 16 + m) << s) + g];
        if (SCALED) v = gl::mul(v, sc[r0 + 4 + s]);
        lds_twB[m * 16 + i] = v;
    }
    __syncthreads();

    const unsigned col0 = blockIdx.y * a.cols_per_block;
    const unsigned col1 = min(col0 + a.cols_per_block, a.n_cols);
    const u32 ta = t >> 4, tc = t & 15;
    // Software pipeline over the columns of this workgroup: VMEM returns in issue order and ONE counter (vmcnt) covers loads and
    // stores, so a column's loads queued BEHIND the previous column's stores cannot be consumed before those stores have been
    // acknowledged by the memory system (A/B builds: without the stores the pass is 23 % faster, without the loads 17 %,
    // tools/ntt_wait_ab.sh).  Here the next column's 16 words are requested right after the data registers of the current one have
    // gone to LDS for the last transpose, AHEAD of its stores: the wait at the top of the next iteration is vmcnt(16) — the stores
    // stay in flight under the butterflies — and the load latency hides under the store phase.
    auto request = [&](u64 (&v)[16], unsigned col) {
        const gcptr src = uniform_gptr(a.in + (size_t)col * a.in_col_stride + (size_t)coset * a.in_coset_stride + (size_t)b * TILE);
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = BJ_R16_LD(src + j * 256, t * 8u, j + col);
    };
    // The first column is peeled off the loop: the register scoreboard of the compiler is merged at a loop header, and with the
    // prologue's loads (nothing behind them) on one edge and the latch's loads (sixteen stores behind them) on the other it would
    // wait for vmcnt(0) at the top of every iteration — stores included.  With the peeled copy both edges look alike.
    u64 x[16];
    auto column = [&](unsigned col) {
        const gptr dst = (gptr)uniform_gptr(a.out + (size_t)col * a.out_col_stride + (size_t)coset * n + (size_t)b * TILE);
        if (!BJ_R16_PREFETCH_LOCAL) request(x, col);
        radix16<false, 12 - ROUNDS>(x, lds_tw);   // step A: bits 11..8 in registers, twiddles uniform (LDS broadcasts at the point of use)
#pragma unroll
        for (int j = 0; j < 16; j++) lds[pad(j * 256 + t)] = x[j];
        BJ_R16_SYNC_L();
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = lds[pad(ta * 256 + j * 16 + tc)];
        BJ_R16_SYNC_L();
        radix16_lds<false>(x, lds_twB + ta * 16);   // step B: bits 7..4
#pragma unroll
        for (int j = 0; j < 16; j++) lds[pad(ta * 256 + j * 16 + tc)] = x[j];
        BJ_R16_SYNC_L();
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = lds[pad(t * 16 + j)];
        BJ_R16_SYNC_L();
        radix16<false>(x, twC);   // step C: bits 3..0
#pragma unroll
        for (int j = 0; j < 16; j++) lds[pad(t * 16 + j)] = gl::canon(x[j]);   // the transform's output: canonical residues
        BJ_R16_SYNC_L();
        if (BJ_R16_PREFETCH_LOCAL && col + 1 < col1) request(x, col + 1);            // x is free: its values sit in LDS
        if (BJ_R16_WIDE_ST) {   // lane t: words 2u, 2u + 1 (u = t & 127) of row 2k + (t >> 7): one 16-byte store, 1 KB per wave instruction
            const u32 u2 = (t & 127u) * 2u, rh = t >> 7;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const u32 l = (2u * k + rh) * 256u + u2;
                BJ_R16_ST_GUARD(a) st_off2(dst, l * 8u, lds[pad(l)], lds[pad(l + 1)]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; j++) BJ_R16_ST_GUARD(a) st_off(dst + j * 256, t * 8u, lds[pad(j * 256 + t)]);
        }
        BJ_R16_SYNC_L();
    };
    if (col0 >= col1) return;
    if (BJ_R16_PREFETCH_LOCAL) request(x, col0);
    column(col0);
#pragma nounroll
    for (unsigned col = col0 + 1; col < col1; col++) column(col);
}

// ---------------------------------------------------------------------------------------------------------
#ifndef BJ_S8_WAVES
#define BJ_S8_WAVES 3   // tuned: 3 -> 7.6 ms, 4 -> 9.7 ms (spills) per 93 x 2^20 x 8 LDE
#endif
template <bool SCALED, bool UNIT_FIRST>
__global__ void __launch_bounds__(256, BJ_S8_WAVES) ntt_strided8_kernel(R16Args a) {
    __shared__ u64 lds[LDS_ELEMS + 16 + 16 * 16];
    u64 *lds_tw = lds + LDS_ELEMS;            // 15 block-uniform twiddles of the first step
    u64 *lds_tw2 = lds_tw + 16;               // [tm][15 (+1 pad)] twiddles of the second step: all twiddles live in LDS, which
                                              // leaves the VGPRs to the 16 data elements and lets four waves share a SIMD
    const u32 t = threadIdx.x;
    const unsigned coset = blockIdx.z;
    const unsigned rem_log = a.log_n - a.r0 - 8;          // bits of "lo" (>= 4)
    const u32 tiles_per_hi = 1u << (rem_log - 4);
    const u32 hi = blockIdx.x / tiles_per_hi, lo_tile = blockIdx.x % tiles_per_hi;
    const size_t n = (size_t)1 << a.log_n;
    const u64 *sc = SCALED ? a.round_scale + (size_t)coset * 32 : nullptr;
    const u32 tm = t >> 4, tl = t & 15;

    stage_uniform_twiddles<SCALED>(lds_tw, a.tw, hi, a.r0, sc);
    if (t < 240) {   // second-step twiddles: T[((hi*16 + m) << s) + g] * sc[r0 + 4 + s] for m < 16
        const u32 m = t / 15, i = t % 15;
        const int s = 31 - __clz(i + 1);
        const int g = (int)(i + 1) - (1 << s);
        u64 v = a.tw[(((size_t)hi * 16 + m) << s) + g];
        if (SCALED) v = gl::mul(v, sc[a.r0 + 4 + s]);
        lds_tw2[m * 16 + i] = v;
    }
    __syncthreads();

    // addressing: a wave-uniform pointer per access (SGPR pair, advanced by scalar adds) + ONE 32-bit per-lane offset for the
    // loads and one for the stores; sixteen 64-bit per-lane addresses would cost 64 VGPRs and a wave out of every SIMD
    const size_t tile_base = ((size_t)hi << (a.log_n - a.r0)) + ((size_t)lo_tile << 4);
    const u32 off_ld = ((tm << rem_log) + tl) * 8u, off_st = (((tm * 16) << rem_log) + tl) * 8u;   // bytes; a column is < 2^32 bytes
    const unsigned col0 = blockIdx.y * a.cols_per_block;
    const unsigned col1 = min(col0 + a.cols_per_block, a.n_cols);
    // the same software pipeline as in ntt_local12_kernel: the next column's loads go out before this column's stores (here the
    // stores leave from the data registers themselves, so the prefetched words take sixteen more register pairs: 121 -> ~153 VGPRs,
    // still three waves per SIMD)
    auto request = [&](u64 (&v)[16], unsigned col) {
        const gcptr src = uniform_gptr(a.in + (size_t)col * a.in_col_stride + (size_t)coset * a.in_coset_stride + tile_base);
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = BJ_R16_LD_S8(src + ((size_t)(j * 16) << rem_log), off_ld, j + col);
    };
    u64 x[16];
    auto column = [&](unsigned col) {
        const gptr dst = (gptr)uniform_gptr(a.out + (size_t)col * a.out_col_stride + (size_t)coset * n + tile_base);
        if (!BJ_R16_PREFETCH_STRIDED) request(x, col);
        radix16_lds<UNIT_FIRST>(x, lds_tw);    // mid bits 7..4
#pragma unroll
        for (int j = 0; j < 16; j++) lds[pad(j * 256 + t)] = x[j];
        BJ_R16_SYNC_S();
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = lds[pad(tm * 256 + j * 16 + tl)];
        BJ_R16_SYNC_S();
        // the next column's words are requested BEFORE the second step: they travel under its butterflies and are in the queue
        // ahead of this column's stores (the copy nx -> x at the bottom then finds them there: vmcnt(16), the stores stay in flight)
        u64 nx[16];
        const bool more = BJ_R16_PREFETCH_STRIDED && col + 1 < col1;   // wave-uniform
        if (more) request(nx, col + 1);
        radix16_lds<false>(x, lds_tw2 + tm * 16);   // mid bits 3..0
#pragma unroll
        for (int j = 0; j < 16; j++) BJ_R16_ST_GUARD_S8(a) st_off(dst + ((size_t)j << rem_log), off_st, x[j]);
        if (more) {
#pragma unroll
            for (int j = 0; j < 16; j++) x[j] = nx[j];
        }
    };
    if (col0 >= col1) return;
    if (BJ_R16_PREFETCH_STRIDED) request(x, col0);
    column(col0);          // peeled (see ntt_local12_kernel): both edges into the loop carry loads with sixteen stores behind them
#pragma nounroll
    for (unsigned col = col0 + 1; col < col1; col++) column(col);
}

// ---------------------------------------------------------------------------------------------------------
template <bool SCALED, bool UNIT_FIRST>
__global__ void __launch_bounds__(256) ntt_strided4_kernel(R16Args a) {
    __shared__ u64 lds_tw[16];
    const u32 t = threadIdx.x;
    const unsigned coset = blockIdx.z;
    const unsigned rem_log = a.log_n - a.r0 - 4;          // >= 8
    const u32 tiles_per_hi = 1u << (rem_log - 8);
    const u32 hi = blockIdx.x / tiles_per_hi, lo_tile = blockIdx.x % tiles_per_hi;
    const size_t n = (size_t)1 << a.log_n;
    const u64 *sc = SCALED ? a.round_scale + (size_t)coset * 32 : nullptr;
    stage_uniform_twiddles<SCALED>(lds_tw, a.tw, hi, a.r0, sc);
    __syncthreads();
    u64 tw1[15];
#pragma unroll
    for (int i = 0; i < 15; i++) tw1[i] = lds_tw[i];
    const size_t base = ((size_t)hi << (a.log_n - a.r0)) + ((size_t)lo_tile << 8);
    const unsigned col0 = blockIdx.y * a.cols_per_block;
    const unsigned col1 = min(col0 + a.cols_per_block, a.n_cols);
    for (unsigned col = col0; col < col1; col++) {
        const u64 *src = a.in + (size_t)col * a.in_col_stride + (size_t)coset * a.in_coset_stride + base;
        u64 *dst = a.out + (size_t)col * a.out_col_stride + (size_t)coset * n + base;
        u64 x[16];
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = ld_off(src + ((size_t)j << rem_log), t * 8u);
        radix16<UNIT_FIRST>(x, tw1);
#pragma unroll
        for (int j = 0; j < 16; j++) st_off(dst + ((size_t)j << rem_log), t * 8u, x[j]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------
// The FIRST four rounds (strides n/2 .. n/16) of a transform with 14 + 4k rounds, for every coset at once: a lane owns the 16
// elements {i + m * n/16} of two adjacent indices i (16-byte accesses, 1 KB per wave instruction), reads them ONCE and emits
// every coset from them (an LDE reads its monomials once per column, not once per coset).  The 15 twiddles T[g] * sc[r] of
// these rounds are uniform per coset and live in LDS.  No tile and no barrier in the data path.  The pass is bound by its
// 8 * (1 + cosets) * n bytes per column, so its butterflies ride on VALU slots that would idle anyway: the local pass behind
// it then runs 10 rounds instead of 12 (ntt_local12_kernel<., 10>).
// HOIST = true keeps the 16 (x V) input words of a lane in registers across the coset loop (206 VGPRs with V = 2: two waves per
// SIMD, whose butterflies and stores then run one after the other: the pass took the SUM of its VALU time and its HBM time).
// HOIST = false asks for them again for every coset through an opaque pointer — the seven repeats hit the L2 (a workgroup's
// inputs are 64 KB) — so the register file holds one working set and four waves share a SIMD: stores of one wave drain under
// the butterflies of the others.
template <bool SCALED, int V, int WAVES /* 0: HOIST; else reload with this many waves per SIMD asked of the register allocator */>
__global__ void __launch_bounds__(256, WAVES ? WAVES : 1) ntt_first4_kernel(R16Args a, unsigned n_cosets) {
    constexpr bool HOIST = WAVES == 0;
    __shared__ u64 tws[64 * 16];                          // [coset][15 (+1 pad)]
    const size_t n = (size_t)1 << a.log_n, sl = n >> 4;
    for (u32 t = threadIdx.x; t < n_cosets * 15; t += blockDim.x) {
        const u32 c = t / 15, i = t % 15;
        const int r = 31 - __clz(i + 1);
        const int g = (int)(i + 1) - (1 << r);
        u64 v = a.tw[g];
        if (SCALED) v = gl::mul(v, a.round_scale[(size_t)c * 32 + r]);
        tws[c * 16 + i] = v;
    }
    __syncthreads();
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (i >= sl) return;
    const u64 *src = a.in + (size_t)blockIdx.y * a.in_col_stride + i;
    u64 *dst = a.out + (size_t)blockIdx.y * a.out_col_stride + i;
    auto load = [](const u64 *p, u64 &lo, u64 &hi) {
        if (V == 2) {
            const ulonglong2 q = *reinterpret_cast<const ulonglong2 *>(p);
            lo = q.x;
            hi = q.y;
        } else {
            lo = p[0];
        }
    };
    const bool shared_input = a.in_coset_stride == 0;
    u64 in0[HOIST ? 16 : 1], in1[HOIST && V == 2 ? 16 : 1];
    if (HOIST && shared_input) {
#pragma unroll
        for (int m = 0; m < 16; m++) load(src + (size_t)m * sl, in0[m], in1[V == 2 ? m : 0]);
    }
    for (unsigned c = 0; c < n_cosets; c++) {
        u64 x0[16], x1[V == 2 ? 16 : 1];
        const u64 *cs = src + (size_t)c * a.in_coset_stride;
        if (!HOIST) asm volatile("" : "+v"(cs));          // a fresh pointer per coset: the loads below stay inside the loop
#pragma unroll
        for (int m = 0; m < 16; m++) {
            if (HOIST && shared_input) {
                x0[m] = in0[m];
                if (V == 2) x1[m] = in1[m];
            } else {
                load(cs + (size_t)m * sl, x0[m], x1[V == 2 ? m : 0]);
            }
        }
        const u64 *tw = tws + c * 16;
        radix16<!SCALED>(x0, tw);   // unscaled: T[0] = 1, the first round is additions only
        if constexpr (V == 2) radix16<!SCALED>(x1, tw);
        u64 *o = dst + (size_t)c * n;
#pragma unroll
        for (int m = 0; m < 16; m++) {
            if (V == 2)
                *reinterpret_cast<ulonglong2 *>(o + (size_t)m * sl) = make_ulonglong2(x0[m], x1[V == 2 ? m : 0]);
            else
                o[(size_t)m * sl] = x0[m];
        }
    }
}

// The first FIVE rounds the same way, one index per lane (32 elements {i + m * n/32} in registers): round 0 pairs the two
// halves of the register file, rounds 1..4 are one radix-16 step on each half with that half's twiddles.
template <bool SCALED>
__global__ void __launch_bounds__(256) ntt_first5_kernel(R16Args a, unsigned n_cosets) {
    __shared__ u64 tws[64 * 40];                          // [coset]: [0] round 0; [8 + 16 h + i] step twiddles of half h
    const size_t n = (size_t)1 << a.log_n, sl = n >> 5;
    for (u32 t = threadIdx.x; t < n_cosets * 31; t += blockDim.x) {
        const u32 c = t / 31, i = t % 31;
        const int r = 31 - __clz(i + 1);
        const int g = (int)(i + 1) - (1 << r);
        u64 v = a.tw[g];
        if (SCALED) v = gl::mul(v, a.round_scale[(size_t)c * 32 + r]);
        if (r == 0) {
            tws[c * 40] = v;
        } else {
            const int st = r - 1, h = g >> st, gg = g & ((1 << st) - 1);
            tws[c * 40 + 8 + 16 * h + (1 << st) - 1 + gg] = v;
        }
    }
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sl) return;
    const u64 *src = a.in + (size_t)blockIdx.y * a.in_col_stride + i;
    u64 *dst = a.out + (size_t)blockIdx.y * a.out_col_stride + i;
    u64 in[32];
    if (a.in_coset_stride == 0) {
#pragma unroll
        for (int m = 0; m < 32; m++) in[m] = src[(size_t)m * sl];
    }
    for (unsigned c = 0; c < n_cosets; c++) {
        u64 lo[16], hi[16];
#pragma unroll
        for (int m = 0; m < 16; m++) {
            if (a.in_coset_stride == 0) {
                lo[m] = in[m];
                hi[m] = in[m + 16];
            } else {
                lo[m] = src[(size_t)c * a.in_coset_stride + (size_t)m * sl];
                hi[m] = src[(size_t)c * a.in_coset_stride + (size_t)(m + 16) * sl];
            }
        }
        const u64 *tw = tws + c * 40;
        const u64 w0 = tw[0];
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            if (SCALED)
                gl::butterfly2_weak(lo[j], hi[j], w0, lo[j + 1], hi[j + 1], w0);
            else
                gl::addsub2_weak(lo[j], hi[j], lo[j + 1], hi[j + 1]);      // T[0] = 1
        }
        radix16<false>(lo, tw + 8);
        radix16<false>(hi, tw + 24);
        u64 *o = dst + (size_t)c * n;
#pragma unroll
        for (int m = 0; m < 16; m++) {
            o[(size_t)m * sl] = lo[m];
            o[(size_t)(m + 16) * sl] = hi[m];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// The first TEN rounds (strides n/2 .. n/1024) of a 2^22-point transform in ONE pass over every coset: with ntt_local12 behind it
// an LDE'd column makes two HBM passes (25 n words) instead of the three of first4 + strided8 + local10 (41 n), an inverse
// transform two instead of three.  Tile = 1024 "mid" indices (the ten high bits of the element index) x 8 consecutive "lo"
// indices: 8192 elements = 64 KB of LDS, 512 threads with 16 elements each, two workgroups per CU (<= 128 VGPRs: four waves per
// SIMD), every HBM access a 64-byte run (the two tiles that share a 128-byte line are given to the same XCD back to back —
// blocks b and b + 8 — so that the halves meet in that XCD's L2).  Three register steps per coset: radix-16 on mid bits 9..6,
// radix-16 on bits 5..2, radix-4 on bits 1..0 (four lo values per lane, so that a lane stores 32 contiguous bytes), with two
// transposes through LDS in between.  LDS indices are XOR-swizzled instead of padded (tools/lds_conflicts.py: every ds_read_b64 /
// ds_write_b64 below is conflict-free), which keeps the tile at exactly 64 KB.
// Twiddles: round r < 10 of group k < 2^r needs T[k] * sc_c[r]; all of them are uniform over the tiles, so they come from a
// table tw[coset][2^r - 1 + k] (1023 entries per coset, front10_table_kernel) instead of being rebuilt by every workgroup:
// step A's fifteen per coset sit in LDS for the whole kernel, step B's 16 x 15 are re-staged per coset, step C's three per lane
// are loaded into registers.  The inputs of a tile are read again for every coset (no room for them in 128 VGPRs): seven of the
// eight reads are L2 / Infinity-Cache hits.
struct F10Args {
    const u64 *in;
    u64 *out;
    const u64 *tw;            // [n_cosets][1024]
    unsigned log_n, n_cols, cols_per_block, n_cosets;
    size_t in_col_stride, out_col_stride;
};
// TILED layout of a 2^22-word column (the monomial form between an inverse transform and the extensions that read it):
// element e = m * 4096 + r * 2^LB + l (m: ten bits, r: 12 - LB, l: LB = TILED_LB bits) sits at word
//     tiled(e) = r * (1024 * 2^LB) + (l >> 1) * 2048 + m * 2 + (l & 1)
// — the 1024 * 2^LB words one front-pass tile reads are contiguous, ordered so that (a) the pair of 4096-word chunks (b, b + 512) of the
// inverse transform's last pass, whose results are the words (m, l = 2k) and (m, l = 2k + 1) of FOUR tiles for every m, leaves them
// as 16-byte words in runs of 1 KB per wave store — the bit reversal of ifft_natural_to_natural (fft/mod.rs:464-491) happens in the
// store addresses, no pass of its own — and (b) the front pass fetches 16-byte words (m, l pair) for its LDS tile in [m][l] order as before.
#ifndef BJ_TILED_LB
#define BJ_TILED_LB 3   // lo values per front-pass tile of the tiled layout = 2^LB.  3: 1024 x 8 tiles (64 KB), 512-thread workgroups, TWO per CU —
                        // their barrier phases interleave; the half-line stores of tiles 2u, 2u + 1 meet in one XCD's L2 (blocks b, b + 8).
                        // Measured with tiled (contiguous) reads: LDE 93 x 2^22 x 8 27.10 ms against 27.76 ms for 4 (full-line tiles, one
                        // 1024-thread workgroup per CU); with natural-order reads (64-byte runs) the half-line tiles had lost.
#endif
constexpr int TILED_LB = BJ_TILED_LB;
__host__ __device__ constexpr size_t tiled_index(size_t e) {   // e = m * 4096 + r * 2^LB + l
    return ((e & 4095u) >> TILED_LB) * ((size_t)1024 << TILED_LB) + ((e & ((1u << TILED_LB) - 1u)) >> 1) * 2048u + (e >> 12) * 2u + (e & 1u);
}
__global__ void front10_table_kernel(u64 *out, const u64 *__restrict__ T, const u64 *__restrict__ round_scale, unsigned n_cosets) {
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_cosets * 1024u) return;
    const u32 c = idx >> 10, e = idx & 1023u;
    if (e == 1023u) {
        out[idx] = 0;
        return;
    }
    const int r = 31 - __clz(e + 1);
    const u32 k = e + 1 - (1u << r);
    u64 v = T[k];
    if (round_scale) v = gl::mul(v, round_scale[(size_t)c * 32 + r]);
    out[idx] = gl::canon(v);
}
// The B -> C index map is LINEAR over GF(2) in the bits of (lane, register): index(lane, i) = index(lane, 0) ^ index(0, i), so a lane
// keeps ONE value for it and reaches register i's slot with one v_xor by a literal (padding would need sixteen live addresses).
// It permutes the low five index bits only: the 1024 elements of a wave's region stay inside it.  LB = log2(lo values per tile).
template <int LB>
__host__ __device__ constexpr u32 f10_bc(u32 m, u32 l) {
    const u32 x = (m << LB) + l;
    return LB == 3 ? x ^ (((x >> 3) ^ (x >> 5)) & 31u) : x ^ (((x >> 2) ^ (x >> 4)) & 31u);
}
__device__ __forceinline__ u64 &f10_at(u64 *lds, u32 byte_index) { return *reinterpret_cast<u64 *>(reinterpret_cast<char *>(lds) + byte_index); }
// four rounds on 16 registers, stage s reading its 2^s twiddles at tw[off_s + g] (one table for all ten rounds of a coset)
template <bool UNIT_FIRST, bool PRIO>
__device__ __forceinline__ void radix16_tab(u64 (&x)[16], const u64 *tw, u32 o0, u32 o1, u32 o2, u32 o3) {
    const u32 off[4] = {o0, o1, o2, o3};
#pragma unroll
    for (int s = 0; s < 3; s++) {
        if (PRIO) set_prio(3 - s);
        const int half = 8 >> s;
#pragma unroll
        for (int g = 0; g < (1 << s); g++) {
            const u64 w = (UNIT_FIRST && s == 0) ? 1 : tw[off[s] + g];
#pragma unroll
            for (int j = 0; j < half; j += 2) {
                const int iu = g * 2 * half + j, iv = iu + half;
                if (UNIT_FIRST && s == 0)
                    gl::addsub2_weak(x[iu], x[iv], x[iu + 1], x[iv + 1]);
                else
                    gl::butterfly2_weak(x[iu], x[iv], w, x[iu + 1], x[iv + 1], w);
            }
        }
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int g = 0; g < 8; g += 2)
        gl::butterfly2_weak(x[2 * g], x[2 * g + 1], tw[off[3] + g], x[2 * g + 2], x[2 * g + 3], tw[off[3] + g + 1]);
}
typedef const void __attribute__((address_space(1))) *f10_gsrc;
typedef void __attribute__((address_space(3))) *f10_ldst;
// A workgroup (8 waves) owns ONE (tile, coset) and loops over columns.  The eight cosets of a tile are eight workgroups that the
// dispatcher places on ONE XCD at the same time (block b runs on XCD b % 8: b = ((window * n_cosets + coset) * 2 + half) * 8 + xcd), so the
// tile crosses the fabric once and the other seven reads hit that XCD's L2 (measured with the cosets looped INSIDE a workgroup
// instead: FETCH_SIZE = 8 x the input, the pass ran at 4 TB/s of L2-miss traffic).  Data flow of one column:
//   the tile arrives by LDS-DMA (global_load_lds_dwordx4: no destination registers), wave w fetching the eight 1-KB pieces of
//   region w = mid bits 9..7; it is requested BEFORE the previous column's eight stores, so the counted wait at the top
//   (vmcnt(8): VMEM operations retire in order) leaves those stores in flight;                                    [barrier 1]
//   step A (mid bits 9..6 in registers) reads and writes the tile in place, in the linear order the DMA left;   [barrier 2]
//   steps B (bits 5..2) and C (bits 1..0 x four lo values) of wave w touch region w only — the B -> C transpose is an exchange
//   among the lanes of ONE wave, ordered by the wave's own LDS queue, no barrier — and when wave w has read its step-C
//   registers, region w is dead: it requests the next column's pieces at once, then computes step C and stores.
// The coset's twiddle table (8 KB) is fetched once per workgroup, by LDS-DMA as well.
#ifndef BJ_F10_PRIO
#define BJ_F10_PRIO 1
#endif
template <bool UNIT_FIRST, int LB, bool TILED_IN = false>
__global__ void __launch_bounds__(64 << LB, 4) ntt_front10_kernel(F10Args a) {
    static_assert(!TILED_IN || LB == TILED_LB, "the tiled layout is defined for one tile width");
    constexpr u32 NT = 64u << LB, TILE_E = 1024u << LB, ROWS = 128u >> LB;   // threads, elements per tile, mid rows per 1-KB DMA piece
    __shared__ u64 lds[TILE_E + 1024];      // ONE object: tile (64 / 128 KB), twiddle table of this workgroup's coset (8 KB)
    u64 *lds_tw = lds + TILE_E;
    const u32 t = threadIdx.x;
    const u32 wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63u;
    const unsigned s_log = a.log_n - 10;
    const size_t n = (size_t)1 << a.log_n;
    const u32 b = blockIdx.x;
    u32 c, tile;
    if (LB == 3) {   // tiles 2u, 2u + 1 (the halves of 128-byte lines): blocks b, b + 8 — one XCD, back to back
        const u32 xcd = b & 7u, half = (b >> 3) & 1u, window = (b >> 4) / a.n_cosets;
        c = (b >> 4) % a.n_cosets;
        tile = window * 16u + xcd * 2u + half;
    } else {
        const u32 xcd = b & 7u, window = (b >> 3) / a.n_cosets;
        c = (b >> 3) % a.n_cosets;
        tile = window * 8u + xcd;
    }
    const size_t lo0 = (size_t)tile << LB;
    // step A: lane = (ml : mid bits 5..0, l); registers = mid bits 9..6: LDS index i * NT + t (as the DMA leaves the tile)
    // step B: lane = (mh : mid bits 9..6, mll : bits 1..0, l); registers = bits 5..2
    const u32 lB = t & ((1u << LB) - 1u), mllB = (t >> LB) & 3u, mhB = t >> (LB + 2);
    // step C: lane = (m92 : mid bits 9..2, lhi : lo bits above 1..0); registers = (mid bits 1..0, lo bits 1..0)
    const u32 lhiC = t & ((1u << (LB - 2)) - 1u), m92 = t >> (LB - 2);
    const u32 offC = (((m92 * 4u) << s_log) + lhiC * 4u) * 8u;
    const u32 ixB1 = (((mhB * 64u + mllB) << LB) + lB) * 8u, ixB2 = f10_bc<LB>(mhB * 64u + mllB, lB) * 8u, ixC = f10_bc<LB>(m92 * 4u, lhiC * 4u) * 8u;
    // DMA source of this lane inside a piece: piece p = mid rows [ROWS p, ROWS (p + 1)), lane = (row, lo pair)
    // (tiled input: the lane's 16 bytes are the words (row, l pair) at pair * 2048 + row * 2 of the tile's 1024 * 2^LB contiguous words)
    const u32 dma_off = TILED_IN ? ((lane & ((1u << (LB - 1)) - 1u)) * 2048u + (lane >> (LB - 1)) * 2u) * 8u
                                 : (((lane >> (LB - 1)) << s_log) + (lane & ((1u << (LB - 1)) - 1u)) * 2u) * 8u;
    const unsigned col0 = blockIdx.y * a.cols_per_block;
    const unsigned col1 = min(col0 + a.cols_per_block, a.n_cols);
    if (col0 >= col1) return;
    auto request_tile = [&](unsigned col) {
        const char __attribute__((address_space(1))) *src =
            (const char __attribute__((address_space(1))) *)uniform_gptr(a.in + (size_t)col * a.in_col_stride + (TILED_IN ? (size_t)tile * TILE_E : lo0)) + dma_off;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const u32 piece = wave * 8u + k;
            const size_t piece_off = TILED_IN ? (size_t)piece * ROWS * 16u : (((size_t)piece * ROWS) << s_log) * 8u;
            __builtin_amdgcn_global_load_lds((f10_gsrc)(src + piece_off), (f10_ldst)(lds + piece * 128u), 16, 0, 0);
        }
    };
    if (wave < 8u) {
        const char __attribute__((address_space(1))) *tsrc =
            (const char __attribute__((address_space(1))) *)uniform_gptr(a.tw + (size_t)c * 1024u + wave * 128u) + lane * 16u;
        __builtin_amdgcn_global_load_lds((f10_gsrc)tsrc, (f10_ldst)(lds_tw + wave * 128u), 16, 0, 0);
    }
    request_tile(col0);
    const gptr dst0 = (gptr)uniform_gptr(a.out + (size_t)c * n + lo0);
    for (unsigned col = col0; col < col1; col++) {
        // opaque copies: the sixteen xor-ed indices of an arrangement are formed where they are used, not hoisted out of the loop
        u32 jB1 = ixB1, jB2 = ixB2, jC = ixC;
        asm volatile("" : "+v"(jB1), "+v"(jB2), "+v"(jC));
        if (col == col0)
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        else
#ifdef BJ_F10_AB_NOWAIT   // timing experiment (wrong results): what the pass would cost if the tile were always there
            asm volatile("s_barrier" ::: "memory");
#else
            asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");   // this column's pieces have landed; the last eight stores may still fly
#endif
        u64 x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] = lds[i * NT + t];
        radix16<UNIT_FIRST, 0, BJ_F10_PRIO>(x, lds_tw);
#pragma unroll
        for (int i = 0; i < 16; i++) lds[i * NT + t] = x[i];          // in place: no barrier between the read and this write
#ifdef BJ_F10_AB_NOBAR2
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
        BJ_R16_SYNC_LDS();
#endif
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] = f10_at(lds, jB1 + i * (32u << LB));
        radix16_tab<false, BJ_F10_PRIO>(x, lds_tw, 15u + mhB, 31u + 2u * mhB, 63u + 4u * mhB, 127u + 8u * mhB);
#pragma unroll
        for (int i = 0; i < 16; i++) f10_at(lds, jB2 ^ (f10_bc<LB>(i * 4u, 0) * 8u)) = x[i];      // inside this wave's region
        const u64 w8 = lds_tw[255u + m92], w9a = lds_tw[511u + 2u * m92], w9b = lds_tw[512u + 2u * m92];
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] = f10_at(lds, jC ^ (f10_bc<LB>(i >> 2, i & 3) * 8u));     // written by lanes of this wave
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of its region are complete: the region is dead
        if (col + 1 < col1) request_tile(col + 1);
        // round 8: mid bit 1 (registers 8 apart), group m92; round 9: mid bit 0 (registers 4 apart), groups 2 m92, 2 m92 + 1
        if (BJ_F10_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ll = 0; ll < 4; ll += 2) {
            gl::butterfly2_weak(x[ll], x[8 + ll], w8, x[4 + ll], x[12 + ll], w8);
            gl::butterfly2_weak(x[ll + 1], x[9 + ll], w8, x[5 + ll], x[13 + ll], w8);
        }
        if (BJ_F10_PRIO) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int ll = 0; ll < 4; ll++) gl::butterfly2_weak(x[ll], x[4 + ll], w9a, x[8 + ll], x[12 + ll], w9b);
        const gptr dst = dst0 + (size_t)col * a.out_col_stride;
#pragma unroll
        for (int mm = 0; mm < 4; mm++) {
            st_off2(dst + ((size_t)mm << s_log), offC, x[4 * mm], x[4 * mm + 1]);
            st_off2(dst + ((size_t)mm << s_log), offC + 16u, x[4 * mm + 2], x[4 * mm + 3]);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// LAST pass of a 2^22-point INVERSE transform whose result is kept in the tiled layout: the twelve local rounds of ntt_local12 on the
// chunk pair (b, b + 512) — one after the other through one LDS tile — the factor 1 / n, and the store.  After step C lane t holds
// the words p = 16 t + j of a chunk; their natural index is bitrev22(4096 b + p) = m * 4096 + r * 16 + l with
//     m = rev10(p & 1023) = rev4(j) * 64 + rev6(t & 63),   r = rev2(p >> 10) * 64 + (rev10(b) >> 4),   l = rev10(b) & 15,
// and rev10(b + 512) = rev10(b) + 1: the two chunks are the two halves of every 16-byte word (m, l pair) of the tiled layout, and
// for a fixed register j the 64 lanes of a wave cover 64 consecutive m — one wave store = 1 KB contiguous, straight from the
// registers (ntt_local12's third transpose through LDS, there only for the stores' sake, is not needed here).  Step C is free to
// choose WHICH sixteen-word group a lane takes: lane t takes group pi(t) = t with its low six bits reversed, so that m = rev4(j) * 64
// + (t & 63) and the lanes of a store are in address order (in lane order rev6 the same kilobyte leaves as 64 separate 16-byte
// writes: measured, the pass is then bound by them).  The twiddles of steps
// A and B of both chunks stay in LDS (2 x (16 + 256) words); step C's fifteen per lane are fetched again for every chunk of every
// column (L2 hits, issued with the chunk's sixteen data loads) — resident for both chunks they cost 60 VGPRs and the third wave of a SIMD.
struct PairArgs {
    const u64 *in;
    u64 *out;
    const u64 *tw;           // bit-reversed twiddle table of the inverse root
    const u64 *tw_scaled;    // the same 2^21 entries times 1 / n (the last round's twiddles: radix16_scaled)
    u64 scale;               // 1 / n
    unsigned n_cols, cols_per_block;
    size_t in_col_stride, out_col_stride;
};
#ifndef BJ_PAIR_WAVES
#define BJ_PAIR_WAVES 3
#endif
__global__ void __launch_bounds__(256, BJ_PAIR_WAVES) ntt_local12_pair_tiled_kernel(PairArgs a) {
    __shared__ u64 lds[LDS_ELEMS + 2 * 16 + 2 * 16 * 16];
    u64 *lds_twA = lds + LDS_ELEMS;           // [chunk][15 (+1)]
    u64 *lds_twB = lds_twA + 32;              // [chunk][t >> 4][15 (+1)]
    const u32 t = threadIdx.x;
    const u32 bp = blockIdx.x;                // chunks bp and bp + 512
    constexpr unsigned r0 = 10;
    (void)r0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const u32 b = bp + 512u * h;
        if (t < 15) {
            const int s = 31 - __clz(t + 1);
            const int g = (int)(t + 1) - (1 << s);
            lds_twA[h * 16 + t] = a.tw[((size_t)b << s) + g];
        }
        if (t < 240) {
            const u32 m = t / 15, i = t % 15;
            const int s = 31 - __clz(i + 1);
            const int g = (int)(i + 1) - (1 << s);
            lds_twB[h * 256 + m * 16 + i] = a.tw[(((size_t)b * 16 + m) << s) + g];
        }
    }
    __syncthreads();
    const unsigned col0 = blockIdx.y * a.cols_per_block;
    const unsigned col1 = min(col0 + a.cols_per_block, a.n_cols);
    const u32 ta = t >> 4, tc = t & 15;
    const u32 wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const u32 rb = gl::bitrev32(bp, 10);                      // even: bp < 512
    const u32 rl = gl::bitrev32(wave, 2) * 1024u + rb;        // the twelve bits (r, l) below m of this wave's natural indices (l & 1 = 0)
    const size_t tile_off = (size_t)(rl >> TILED_LB) * ((size_t)1024 << TILED_LB) + (size_t)((rl & ((1u << TILED_LB) - 1u)) >> 1) * 2048u;
    const u32 pi = (t & ~63u) | gl::bitrev32(t & 63u, 6);     // the group of sixteen words this lane takes in step C
    const u32 off_st = (t & 63u) * 16u;                       // bytes: word pair m = lane (+ rev4(j) * 64, a constant per store)
    for (unsigned col = col0; col < col1; col++) {
        u64 y[16];
#pragma nounroll   // ONE copy of the twelve rounds serves both chunks (unrolled, the loop body is ~50 KB of code for a 64 KB instruction cache)
        for (int h = 0; h < 2; h++) {
            const gcptr src = uniform_gptr(a.in + (size_t)col * a.in_col_stride + (size_t)(bp + 512u * h) * TILE);
            u64 x[16];
#pragma unroll
            for (int j = 0; j < 16; j++) x[j] = ld_off(src + j * 256, t * 8u);
            // step C's twiddles of this chunk: T[(kb << s) + g], kb = 256 b + t; the last round's from the scaled table
            u64 twC[15];
            {
                u32 kb = (bp + 512u * h) * 256u + pi;
                asm volatile("" : "+v"(kb));          // a fresh index per chunk and column: the loads stay inside the loop
#pragma unroll
                for (int st = 0; st < 4; st++)
#pragma unroll
                    for (int g = 0; g < (1 << st); g++) twC[(1 << st) - 1 + g] = (st == 3 ? a.tw_scaled : a.tw)[((size_t)kb << st) + g];
            }
            radix16<false>(x, lds_twA + h * 16);
#pragma unroll
            for (int j = 0; j < 16; j++) lds[pad(j * 256 + t)] = x[j];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; j++) x[j] = lds[pad(ta * 256 + j * 16 + tc)];
            __syncthreads();
            radix16_lds<false>(x, lds_twB + h * 256 + ta * 16);
#pragma unroll
            for (int j = 0; j < 16; j++) lds[pad(ta * 256 + j * 16 + tc)] = x[j];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; j++) x[j] = lds[pad(pi * 16 + j)];
            __syncthreads();
            radix16_scaled(x, twC, a.scale);
            if (h == 0) {
#pragma unroll
                for (int j = 0; j < 16; j++) y[j] = gl::canon(x[j]);
            } else {
                const gptr dst = (gptr)uniform_gptr(a.out + (size_t)col * a.out_col_stride + tile_off);
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    constexpr u32 R4[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
                    st_off2(dst + (size_t)R4[j] * 128u, off_st, y[j], gl::canon(x[j]));
                }
            }
        }
    }
}

// natural <-> tiled re-layout of 2^22-word columns (the quotient's chunks, whose monomials come out of a transform of another size,
// and the operator-level entry points): block = (tile r, 64 consecutive m), 1024 words through LDS; natural side 64 runs of 128
// bytes, tiled side 8 runs of 1 KB
__global__ void __launch_bounds__(256) tiled_permute_kernel(const u64 *in, u64 *out, size_t in_col_stride, size_t out_col_stride, int to_tiled) {
    constexpr u32 LW = 1u << TILED_LB, PER = 64u * LW, TILE_W = 1024u * LW;   // lo values per row, words per block, words per tile
    __shared__ u64 tile[64 * (LW + 1)];
    const u32 t = threadIdx.x, r = blockIdx.x >> 4, m0 = (blockIdx.x & 15u) * 64u;
    const u64 *src = in + (size_t)blockIdx.y * in_col_stride;
    u64 *dst = out + (size_t)blockIdx.y * out_col_stride;
#pragma unroll
    for (u32 k = 0; k < PER / 256u; k++) {
        const u32 q = t + 256u * k;
        if (to_tiled) {          // natural read: q = (m local, l)
            const u32 ml = q >> TILED_LB, l = q & (LW - 1u);
            tile[ml * (LW + 1) + l] = src[((size_t)(m0 + ml) << 12) + r * LW + l];
        } else {                 // tiled read: q = (pair, m local, l & 1)
            const u32 pair = q >> 7, ml = (q >> 1) & 63u, lb = q & 1u;
            tile[ml * (LW + 1) + pair * 2 + lb] = src[(size_t)r * TILE_W + pair * 2048u + (m0 + ml) * 2u + lb];
        }
    }
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < PER / 256u; k++) {
        const u32 q = t + 256u * k;
        if (to_tiled) {
            const u32 pair = q >> 7, ml = (q >> 1) & 63u, lb = q & 1u;
            dst[(size_t)r * TILE_W + pair * 2048u + (m0 + ml) * 2u + lb] = tile[ml * (LW + 1) + pair * 2 + lb];
        } else {
            const u32 ml = q >> TILED_LB, l = q & (LW - 1u);
            dst[((size_t)(m0 + ml) << 12) + r * LW + l] = tile[ml * (LW + 1) + l];
        }
    }
}

static unsigned pick_cols_per_block(unsigned tiles, unsigned n_cols, unsigned n_cosets) {
    // amortise the per-workgroup twiddle preparation over several columns, but keep >= ~4096 workgroups in flight
#ifndef BJ_R16_CPB
#define BJ_R16_CPB 8
#endif
    unsigned cpb = BJ_R16_CPB;
    while (cpb > 1 && (size_t)tiles * ((n_cols + cpb - 1) / cpb) * n_cosets < 4096) cpb >>= 1;
    return cpb;
}

void launch_ntt_local12(const u64 *in, u64 *out, const u64 *tw, const u64 *round_scale, unsigned log_n,
                        unsigned n_cols, unsigned n_cosets, size_t in_col_stride, size_t in_coset_stride,
                        size_t out_col_stride, unsigned rounds, hipStream_t s) {
    unsigned tiles = 1u << (log_n - 12);
    unsigned cpb = pick_cols_per_block(tiles, n_cols, n_cosets);
    R16Args a{in, out, tw, round_scale, log_n, log_n - 12, n_cols, cpb, in_col_stride, in_coset_stride, out_col_stride};
    dim3 grid(tiles, (n_cols + cpb - 1) / cpb, n_cosets);
    if (rounds == 10) {
        if (round_scale)
            hipLaunchKernelGGL((ntt_local12_kernel<true, 10>), grid, dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL((ntt_local12_kernel<false, 10>), grid, dim3(256), 0, s, a);
    } else if (rounds == 9) {
        if (round_scale)
            hipLaunchKernelGGL((ntt_local12_kernel<true, 9>), grid, dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL((ntt_local12_kernel<false, 9>), grid, dim3(256), 0, s, a);
    } else if (round_scale)
        hipLaunchKernelGGL((ntt_local12_kernel<true, 12>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((ntt_local12_kernel<false, 12>), grid, dim3(256), 0, s, a);
}

// last pass of a 2^22-point inverse transform into the tiled layout (in: the front pass's output, chunk b at b * 4096)
__global__ void scale_table_kernel(const u64 *in, u64 *out, size_t count, u64 s) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = gl::mul(in[i], s);
}
void launch_scale_table(const u64 *in, u64 *out, size_t count, u64 scale, hipStream_t s) {
    hipLaunchKernelGGL(scale_table_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, in, out, count, scale);
}
void launch_ntt_local12_pair_tiled(const u64 *in, u64 *out, const u64 *tw, const u64 *tw_scaled, u64 scale, unsigned n_cols,
                                   size_t in_col_stride, size_t out_col_stride, hipStream_t s) {
    unsigned cpb = 8;
    while (cpb > 1 && (size_t)512 * ((n_cols + cpb - 1) / cpb) < 2048) cpb >>= 1;
    PairArgs a{in, out, tw, tw_scaled, scale, n_cols, cpb, in_col_stride, out_col_stride};
    hipLaunchKernelGGL(ntt_local12_pair_tiled_kernel, dim3(512, (n_cols + cpb - 1) / cpb), dim3(256), 0, s, a);
}
void launch_tiled_permute(const u64 *in, u64 *out, unsigned n_cols, size_t in_col_stride, size_t out_col_stride, bool to_tiled, hipStream_t s) {
    hipLaunchKernelGGL(tiled_permute_kernel, dim3((4096u >> TILED_LB) * 16, n_cols), dim3(256), 0, s, in, out, in_col_stride, out_col_stride, to_tiled ? 1 : 0);
}

void launch_ntt_strided8(const u64 *in, u64 *out, const u64 *tw, const u64 *round_scale, unsigned log_n, unsigned r0,
                         unsigned n_cols, unsigned n_cosets, size_t in_col_stride, size_t in_coset_stride,
                         size_t out_col_stride, hipStream_t s) {
    unsigned tiles = 1u << (log_n - 12);
    unsigned cpb = pick_cols_per_block(tiles, n_cols, n_cosets);
    R16Args a{in, out, tw, round_scale, log_n, r0, n_cols, cpb, in_col_stride, in_coset_stride, out_col_stride};
    dim3 grid(tiles, (n_cols + cpb - 1) / cpb, n_cosets);
    if (round_scale)
        hipLaunchKernelGGL((ntt_strided8_kernel<true, false>), grid, dim3(256), 0, s, a);
    else if (r0 == 0)
        hipLaunchKernelGGL((ntt_strided8_kernel<false, true>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((ntt_strided8_kernel<false, false>), grid, dim3(256), 0, s, a);
}

void launch_ntt_strided4(const u64 *in, u64 *out, const u64 *tw, const u64 *round_scale, unsigned log_n, unsigned r0,
                         unsigned n_cols, unsigned n_cosets, size_t in_col_stride, size_t in_coset_stride,
                         size_t out_col_stride, hipStream_t s) {
    unsigned tiles = 1u << (log_n - 12);
    unsigned cpb = pick_cols_per_block(tiles, n_cols, n_cosets);
    R16Args a{in, out, tw, round_scale, log_n, r0, n_cols, cpb, in_col_stride, in_coset_stride, out_col_stride};
    dim3 grid(tiles, (n_cols + cpb - 1) / cpb, n_cosets);
    if (round_scale)
        hipLaunchKernelGGL((ntt_strided4_kernel<true, false>), grid, dim3(256), 0, s, a);
    else if (r0 == 0)
        hipLaunchKernelGGL((ntt_strided4_kernel<false, true>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((ntt_strided4_kernel<false, false>), grid, dim3(256), 0, s, a);
}

// first four rounds of all cosets; the caller has checked first4_applicable()
void launch_ntt_first4(const u64 *in, u64 *out, const u64 *tw, const u64 *round_scale, unsigned log_n, unsigned n_cols,
                       unsigned n_cosets, size_t in_col_stride, size_t in_coset_stride, size_t out_col_stride, hipStream_t s) {
    R16Args a{in, out, tw, round_scale, log_n, 0, n_cols, 1, in_col_stride, in_coset_stride, out_col_stride};
    const size_t sl = ((size_t)1 << log_n) >> 4;
    // two adjacent indices per lane (16-byte accesses, 206 VGPRs: 2 waves per SIMD) or one (8-byte accesses, 4 waves)
    const int v = bj::env().ntt_first4_v;
    const int mode = bj::env().ntt_first4_mode;      // 0: inputs hoisted (round 3), 3 / 4: reloaded per coset, waves per SIMD
    dim3 grid((unsigned)((sl / v + 255) / 256), n_cols, 1);
#define BJ_FIRST4(SC, VV, WV) hipLaunchKernelGGL((ntt_first4_kernel<SC, VV, WV>), grid, dim3(256), 0, s, a, n_cosets)
    if (v == 2) {
        if (round_scale) {
            if (mode == 0) BJ_FIRST4(true, 2, 0); else if (mode == 3) BJ_FIRST4(true, 2, 3); else BJ_FIRST4(true, 2, 4);
        } else {
            if (mode == 0) BJ_FIRST4(false, 2, 0); else if (mode == 3) BJ_FIRST4(false, 2, 3); else BJ_FIRST4(false, 2, 4);
        }
    } else if (round_scale)
        BJ_FIRST4(true, 1, 0);
    else
        BJ_FIRST4(false, 1, 0);
#undef BJ_FIRST4
}

void launch_ntt_first5(const u64 *in, u64 *out, const u64 *tw, const u64 *round_scale, unsigned log_n, unsigned n_cols,
                       unsigned n_cosets, size_t in_col_stride, size_t in_coset_stride, size_t out_col_stride, hipStream_t s) {
    R16Args a{in, out, tw, round_scale, log_n, 0, n_cols, 1, in_col_stride, in_coset_stride, out_col_stride};
    const size_t sl = ((size_t)1 << log_n) >> 5;
    dim3 grid((unsigned)((sl + 255) / 256), n_cols, 1);
    if (round_scale)
        hipLaunchKernelGGL(ntt_first5_kernel<true>, grid, dim3(256), 0, s, a, n_cosets);
    else
        hipLaunchKernelGGL(ntt_first5_kernel<false>, grid, dim3(256), 0, s, a, n_cosets);
}

// first ten rounds of all cosets (log_n == 22); d_table: n_cosets * 1024 words of device scratch for the twiddle table
#ifndef BJ_F10_LB
#define BJ_F10_LB 4   // lo values per tile = 2^LB: 3 -> 64-byte runs, 512 threads, two workgroups per CU; 4 -> full lines, 1024 threads, one
#endif
void launch_ntt_front10(const u64 *in, u64 *out, const u64 *tw, const u64 *round_scale, u64 *d_table, unsigned log_n, unsigned n_cols,
                        unsigned n_cosets, size_t in_col_stride, size_t out_col_stride, hipStream_t s, bool tiled_in) {
    hipLaunchKernelGGL(front10_table_kernel, dim3(n_cosets * 4), dim3(256), 0, s, d_table, tw, round_scale, n_cosets);
    constexpr int LBN = BJ_F10_LB;
    const int lb = tiled_in ? TILED_LB : LBN;
    const unsigned tiles = 1u << (log_n - 10 - lb);
    const size_t n = (size_t)1 << log_n;
    for (unsigned c0 = 0; c0 < n_cosets; c0 += 8) {   // at most eight cosets per launch: a tile's workgroups are one dispatch window on one XCD
        const unsigned nc = n_cosets - c0 < 8 ? n_cosets - c0 : 8;
        unsigned cpb = 8;
        while (cpb > 1 && (size_t)tiles * nc * ((n_cols + cpb - 1) / cpb) < 4096) cpb >>= 1;
        F10Args a{in, out + (size_t)c0 * n, d_table + (size_t)c0 * 1024, log_n, n_cols, cpb, nc, in_col_stride, out_col_stride};
        dim3 grid(tiles * nc, (n_cols + cpb - 1) / cpb, 1);
        if (tiled_in) {
            if (round_scale)
                hipLaunchKernelGGL((ntt_front10_kernel<false, TILED_LB, true>), grid, dim3(64u << TILED_LB), 0, s, a);
            else
                hipLaunchKernelGGL((ntt_front10_kernel<true, TILED_LB, true>), grid, dim3(64u << TILED_LB), 0, s, a);
        } else if (round_scale)
            hipLaunchKernelGGL((ntt_front10_kernel<false, LBN>), grid, dim3(64u << LBN), 0, s, a);
        else
            hipLaunchKernelGGL((ntt_front10_kernel<true, LBN>), grid, dim3(64u << LBN), 0, s, a);
    }
}

}  // namespace bj
