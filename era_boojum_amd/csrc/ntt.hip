// Batched radix-2 Goldilocks NTT / coset LDE / iNTT kernels for gfx950 (hand-written HIP, no CUDA shims).
//
// What is computed (must equal, as canonical residues, the reference's CPU path):
//   forward  : fft_natural_to_bitreversed        src/fft/mod.rs:398-411 + serial_ct_ntt :659-734
//   inverse  : ifft_natural_to_natural           src/fft/mod.rs:464-491
//   LDE      : transform_monomials_to_lde        src/cs/implementations/utils.rs:311-403
//   twiddles : precompute_twiddles_for_fft       src/cs/implementations/utils.rs:88-125  (T[j] = w^bitrev(j), j < n/2)
//
// Structure.  The reference's in-place algorithm has round r (0-based) pair elements n/2^(r+1) apart, 2^r groups,
// group k using T[k].  A transform is cut into passes of consecutive rounds [r0, r0+R); within a pass, index
// i = hi * 2^(log_n-r0) + mid * 2^(log_n-r0-R) + lo, only the R "mid" bits interact.  One workgroup owns one tile
// = (hi, 2^Wl consecutive lo) x all 2^R mid, staged in LDS, so every HBM access is a contiguous run of 2^Wl
// elements (Wl = 0 in the last pass, whose tile is 2^R contiguous elements).
//
// Coset evaluation needs no separate "distribute powers" sweep: scaling the input by shift^i is equivalent to
// multiplying the round-r twiddle by shift^(n/2^(r+1)) (the factor shift^(i mod n/2^(r+1)) commutes through the
// butterfly), so the kernel takes a per-round scale table instead.  Same residues, one HBM pass fewer.
#include "gl.h"
#include <cstdint>
#include <cstdlib>
#include "kernels.h"
#include <cstdlib>

using gl::u64;
using gl::u32;

namespace bj {

// ---------------------------------------------------------------------------------------------------------
// twiddle table generation:  T[j] = w^(bitrev_{log_n-1}(j))
// ---------------------------------------------------------------------------------------------------------
__global__ void twiddle_kernel(u64 *out, unsigned log_n, u64 w) {
    size_t half = (size_t)1 << (log_n - 1);
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= half) return;
    u32 e = gl::bitrev32((u32)j, log_n - 1);
    out[j] = gl::pow(w, e);
}

void launch_twiddles(u64 *d_out, unsigned log_n, bool inverse, hipStream_t s) {
    if (log_n == 0) return;
    u64 w = gl::omega(log_n);
    if (inverse) w = gl::inv(w);
    size_t half = (size_t)1 << (log_n - 1);
    unsigned tpb = 256;
    hipLaunchKernelGGL(twiddle_kernel, dim3((unsigned)((half + tpb - 1) / tpb)), dim3(tpb), 0, s, d_out, log_n, w);
}

// ---------------------------------------------------------------------------------------------------------
// generic LDS pass (any R + Wl <= 13).  v0 kernel: correct for every shape, used for small sizes and as the
// fallback; the specialised register-radix kernels below take over for the big shapes.
// ---------------------------------------------------------------------------------------------------------
struct PassArgs {
    const u64 *in;
    u64 *out;
    const u64 *tw;           // bit-reversed twiddle table (>= 2^(r0+R-1) entries)
    const u64 *round_scale;  // [n_cosets][32] per-round twiddle scale, or nullptr (plain subgroup transform)
    unsigned log_n, r0, R, Wl;
    size_t in_col_stride;    // elements between input columns
    size_t in_coset_stride;  // 0 when every coset reads the same input column (first LDE pass), n when in place
    size_t out_col_stride;   // elements between output columns (each column holds n_cosets * n outputs)
};

__global__ void __launch_bounds__(256) ntt_pass_generic_kernel(PassArgs a) {
    extern __shared__ u64 tile[];
    const unsigned R = a.R, Wl = a.Wl;
    const unsigned tile_log = R + Wl;
    const u32 tile_elems = 1u << tile_log;
    const unsigned rem_log = a.log_n - a.r0 - R;        // bits of "lo"
    const u32 tiles_per_hi = 1u << (rem_log - Wl);
    const u32 hi = blockIdx.x / tiles_per_hi;
    const u32 lo_tile = blockIdx.x % tiles_per_hi;
    const size_t n = (size_t)1 << a.log_n;
    const unsigned col = blockIdx.y, coset = blockIdx.z;
    const u64 *src = a.in + (size_t)col * a.in_col_stride + (size_t)coset * a.in_coset_stride;
    u64 *dst = a.out + (size_t)col * a.out_col_stride + (size_t)coset * n;
    const size_t base = ((size_t)hi << (a.log_n - a.r0)) + ((size_t)lo_tile << Wl);
    const u64 *rs = a.round_scale ? a.round_scale + (size_t)coset * 32 : nullptr;

    for (u32 e = threadIdx.x; e < tile_elems; e += blockDim.x) {
        u32 mid = e >> Wl, lo = e & ((1u << Wl) - 1);
        tile[e] = gl::canon(src[base + ((size_t)mid << rem_log) + lo]);
    }
    __syncthreads();
    for (unsigned s = 0; s < R; s++) {
        const unsigned r = a.r0 + s;
        const u64 sc = rs ? rs[r] : 1;
        const unsigned jbits = R - 1 - s;               // bits of "j" inside a group
        for (u32 q = threadIdx.x; q < (tile_elems >> 1); q += blockDim.x) {
            u32 lo = q & ((1u << Wl) - 1);
            u32 qm = q >> Wl;                           // (g, j)
            u32 g = qm >> jbits, j = qm & ((1u << jbits) - 1);
            u32 mid_u = (g << (R - s)) | j;
            u32 iu = (mid_u << Wl) | lo, iv = iu + (1u << (jbits + Wl));
            u32 k = (hi << s) | g;
            u64 u = tile[iu], v = tile[iv];
            if (r != 0 || rs) {                         // round 0 twiddle is 1 unless coset-scaled
                u64 t = a.tw[k];
                if (rs) t = gl::mul(t, sc);
                v = gl::mul(v, t);
            }
            tile[iu] = gl::add(u, v);
            tile[iv] = gl::sub(u, v);
        }
        __syncthreads();
    }
    for (u32 e = threadIdx.x; e < tile_elems; e += blockDim.x) {
        u32 mid = e >> Wl, lo = e & ((1u << Wl) - 1);
        dst[base + ((size_t)mid << rem_log) + lo] = tile[e];
    }
}

// per-round twiddle scale for a coset shift: sc[r] = shift^(n / 2^(r+1)); shifts travel by value (kernarg)
struct ShiftArgs {
    u64 v[64];
};
__global__ void round_scale_kernel(u64 *out, ShiftArgs shifts, unsigned n_cosets, unsigned log_n) {
    unsigned c = blockIdx.x, r = threadIdx.x;
    if (c >= n_cosets || r >= 32) return;
    u64 v = 1;
    if (r < log_n) {
        v = gl::canon(shifts.v[c]);
        for (unsigned i = 0; i < log_n - 1 - r; i++) v = gl::sqr(v);
    }
    out[(size_t)c * 32 + r] = v;
}
void launch_round_scales(u64 *d_out, const u64 *h_shifts, unsigned n_cosets, unsigned log_n, hipStream_t s) {
    ShiftArgs a;
    for (unsigned c = 0; c < 64; c++) a.v[c] = c < n_cosets ? h_shifts[c] : 1;
    hipLaunchKernelGGL(round_scale_kernel, dim3(n_cosets), dim3(32), 0, s, d_out, a, n_cosets, log_n);
}

// ---------------------------------------------------------------------------------------------------------
// Remainder pass: the first R = 1..3 rounds (largest strides) of a transform whose round count is not 12 + 4k.
// A thread owns the 2^R elements {i + m * n/2^R} of one column, reads them ONCE and produces every coset from them
// (an LDE reads its monomials once per column instead of once per coset); all twiddles are uniform per coset
// (T[g] * sc[r], g < 2^r) and are staged in LDS by the first lanes.  No tile, no barrier in the data path: every access
// is a fully coalesced 512-byte wave access.  Replaces the generic LDS pass for this case (4x faster at 2^22).
// ---------------------------------------------------------------------------------------------------------
template <int R, int V>
__global__ void __launch_bounds__(256) ntt_first_rounds_kernel(PassArgs a, unsigned n_cosets) {
    constexpr int E = 1 << R;
    __shared__ u64 tws[64 * (E - 1)];                    // [coset][2^R - 1]: round r, group g at (1 << r) - 1 + g
    const size_t n = (size_t)1 << a.log_n;
    const size_t quarter = n >> R;
    for (unsigned t = threadIdx.x; t < n_cosets * (E - 1); t += blockDim.x) {
        const unsigned c = t / (E - 1), i = t % (E - 1);
        const int r = 31 - __clz(i + 1);
        const int g = (int)(i + 1) - (1 << r);
        u64 v = a.tw[g];
        if (a.round_scale) v = gl::mul(v, a.round_scale[(size_t)c * 32 + r]);
        tws[t] = v;
    }
    __syncthreads();
    // V adjacent indices per lane: 16-byte accesses (1 KB per wave instruction) when the slice allows it
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (i >= quarter) return;
    const unsigned col = blockIdx.y;
    const u64 *src = a.in + (size_t)col * a.in_col_stride + i;
    u64 *dst = a.out + (size_t)col * a.out_col_stride + i;
    auto load = [](const u64 *p, u64 (&o)[V]) {
        if (V == 2) {
            const ulonglong2 q = *reinterpret_cast<const ulonglong2 *>(p);
            o[0] = gl::canon(q.x);
            o[V - 1] = gl::canon(q.y);
        } else {
            o[0] = gl::canon(p[0]);
        }
    };
    u64 in[E][V];
    if (a.in_coset_stride == 0) {
#pragma unroll
        for (int m = 0; m < E; m++) load(src + (size_t)m * quarter, in[m]);
    }
    for (unsigned c = 0; c < n_cosets; c++) {
        u64 x[E][V];
#pragma unroll
        for (int m = 0; m < E; m++) {
            if (a.in_coset_stride == 0) {
#pragma unroll
                for (int v = 0; v < V; v++) x[m][v] = in[m][v];
            } else {
                load(src + (size_t)c * a.in_coset_stride + (size_t)m * quarter, x[m]);
            }
        }
        const u64 *tw = tws + c * (E - 1);
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int half = E >> (r + 1);
#pragma unroll
            for (int g = 0; g < (1 << r); g++) {
                const u64 w = tw[(1 << r) - 1 + g];
#pragma unroll
                for (int j = 0; j < half; j++) {
                    const int iu = g * 2 * half + j, iv = iu + half;
#pragma unroll
                    for (int v = 0; v < V; v++) {
                        u64 u = x[iu][v];
                        u64 t = (g == 0 && !a.round_scale) ? x[iv][v] : gl::mul(x[iv][v], w);   // T[0] = 1
                        x[iu][v] = gl::add(u, t);
                        x[iv][v] = gl::sub(u, t);
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < E; m++) {
            u64 *o = dst + (size_t)c * n + (size_t)m * quarter;
            if (V == 2)
                *reinterpret_cast<ulonglong2 *>(o) = make_ulonglong2(x[m][0], x[m][V - 1]);
            else
                o[0] = x[m][0];
        }
    }
}

static void launch_first_rounds(const u64 *src, u64 *d_out, const u64 *d_tw, const u64 *d_round_scale, unsigned log_n,
                                unsigned R, unsigned n_cols, unsigned n_cosets, size_t src_col_stride,
                                size_t src_coset_stride, size_t out_col_stride, hipStream_t s) {
    PassArgs a{src, d_out, d_tw, d_round_scale, log_n, 0, R, 0, src_col_stride, src_coset_stride, out_col_stride};
    const size_t quarter = ((size_t)1 << log_n) >> R;
    // two indices per lane need 16-byte aligned columns and an even slice
    const bool wide = quarter % 512 == 0 && src_col_stride % 2 == 0 && src_coset_stride % 2 == 0 && out_col_stride % 2 == 0 &&
                      ((uintptr_t)src % 16) == 0 && ((uintptr_t)d_out % 16) == 0 && !bj::env().ntt_first_narrow;
    dim3 grid((unsigned)((quarter / (wide ? 2 : 1) + 255) / 256), n_cols, 1);
    if (wide) {
        if (R == 1)
            hipLaunchKernelGGL((ntt_first_rounds_kernel<1, 2>), grid, dim3(256), 0, s, a, n_cosets);
        else if (R == 2)
            hipLaunchKernelGGL((ntt_first_rounds_kernel<2, 2>), grid, dim3(256), 0, s, a, n_cosets);
        else
            hipLaunchKernelGGL((ntt_first_rounds_kernel<3, 2>), grid, dim3(256), 0, s, a, n_cosets);
    } else if (R == 1)
        hipLaunchKernelGGL((ntt_first_rounds_kernel<1, 1>), grid, dim3(256), 0, s, a, n_cosets);
    else if (R == 2)
        hipLaunchKernelGGL((ntt_first_rounds_kernel<2, 1>), grid, dim3(256), 0, s, a, n_cosets);
    else
        hipLaunchKernelGGL((ntt_first_rounds_kernel<3, 1>), grid, dim3(256), 0, s, a, n_cosets);
}

// Plan: last pass local with up to LOCAL_MAX rounds; earlier rounds in strided passes of <= STRIDED_MAX rounds.
static constexpr unsigned LOCAL_MAX = 12, STRIDED_MAX = 8, TILE_LOG = 12;

static void launch_generic_pass(const u64 *src, u64 *d_out, const u64 *d_tw, const u64 *d_round_scale,
                                unsigned log_n, unsigned r0, unsigned R, unsigned Wl, unsigned n_cols,
                                unsigned n_cosets, size_t src_col_stride, size_t src_coset_stride,
                                size_t out_col_stride, hipStream_t s) {
    PassArgs a{src, d_out, d_tw, d_round_scale, log_n, r0, R, Wl, src_col_stride, src_coset_stride, out_col_stride};
    unsigned tiles = 1u << (log_n - R - Wl);
    size_t lds = ((size_t)8) << (R + Wl);
    unsigned tpb = (1u << (R + Wl)) / 2;
    if (tpb > 256) tpb = 256;
    if (tpb < 64) tpb = 64;
    hipLaunchKernelGGL(ntt_pass_generic_kernel, dim3(tiles, n_cols, n_cosets), dim3(tpb), lds, s, a);
}

static bool force_generic() { return bj::env().ntt_generic; }

// Pass plan.  log_n < 12 (or BJ_NTT_GENERIC=1): generic LDS passes.  Otherwise the last 12 rounds run in
// ntt_local12, the rounds in front of it in radix-16 strided passes of 8 or 4 rounds, and a remainder of 1..3
// rounds (log_n not of the form 12 + 4k) in one generic strided pass at the very front.
bool ntt_two_pass_applies(const u64 *d_in, const u64 *d_out, unsigned log_n, unsigned n_cosets, size_t in_col_stride, size_t out_col_stride) {
    const bool io16 = ((uintptr_t)d_out % 16) == 0 && out_col_stride % 2 == 0 && ((uintptr_t)d_in % 16) == 0 && in_col_stride % 2 == 0;
    return log_n == 22 && n_cosets <= 64 && io16 && bj::env().ntt_two_pass && !force_generic();
}
void launch_ntt_passes(const u64 *d_in, u64 *d_out, const u64 *d_tw, const u64 *d_round_scale, unsigned log_n,
                       unsigned n_cols, unsigned n_cosets, size_t in_col_stride, size_t out_col_stride,
                       hipStream_t s, u64 *d_front_table, bool tiled_in) {
    const size_t n = (size_t)1 << log_n;
    if (log_n == 0) {  // size-1 transform: canonicalising copy
        launch_generic_pass(d_in, d_out, d_tw, nullptr, 0, 0, 0, 0, n_cols, n_cosets, in_col_stride, 0,
                            out_col_stride, s);
        return;
    }
    // the first pass reads the caller's column (shared by all cosets); later passes run in place on d_out
    const u64 *src = d_in;
    size_t src_col_stride = in_col_stride, src_coset_stride = 0;
    unsigned r0 = 0;
    auto advance = [&](unsigned R) {
        r0 += R;
        src = d_out;
        src_col_stride = out_col_stride;
        src_coset_stride = n;
    };
    if (log_n < 12 || force_generic()) {
        unsigned local = log_n < LOCAL_MAX ? log_n : LOCAL_MAX;
        unsigned rest = log_n - local;
        unsigned n_strided = (rest + STRIDED_MAX - 1) / STRIDED_MAX;
        for (unsigned p = 0; p < n_strided; p++) {
            unsigned R = rest / n_strided + (p < rest % n_strided ? 1 : 0);
            unsigned rem_log = log_n - r0 - R;
            unsigned Wl = TILE_LOG - R;
            if (Wl > rem_log) Wl = rem_log;
            launch_generic_pass(src, d_out, d_tw, d_round_scale, log_n, r0, R, Wl, n_cols, n_cosets, src_col_stride,
                                src_coset_stride, out_col_stride, s);
            advance(R);
        }
        launch_generic_pass(src, d_out, d_tw, d_round_scale, log_n, r0, local, 0, n_cols, n_cosets, src_col_stride,
                            src_coset_stride, out_col_stride, s);
        return;
    }
    unsigned front = log_n - 12;
    // 22 rounds in two passes: ten in ntt_front10 (every coset from one tile of the caller's column), twelve in ntt_local12
    // (the front pass moves 16 bytes per lane on both sides: columns on 16-byte boundaries)
    if (d_front_table && ntt_two_pass_applies(d_in, d_out, log_n, n_cosets, in_col_stride, out_col_stride)) {
        launch_ntt_front10(src, d_out, d_tw, d_round_scale, d_front_table, log_n, n_cols, n_cosets, src_col_stride, out_col_stride, s, tiled_in);
        advance(10);
        launch_ntt_local12(src, d_out, d_tw, d_round_scale, log_n, n_cols, n_cosets, src_col_stride, src_coset_stride,
                           out_col_stride, 12, s);
        return;
    }
    // 14 + 4k and 15 + 4k rounds: the coset-expanding front pass is bound by its traffic whatever it computes, so it takes four
    // or five rounds (ntt_first4 / ntt_first5) and the local pass runs ten or nine instead of twelve — the same number of passes,
    // butterflies moved into idle VALU slots.  13 + 4k rounds keep the remainder pass of one round.
    const int front_policy = bj::env().ntt_front;   // 0: remainder passes only (the round-1 plan); 5: first5 wherever it applies; default: measured best
    const bool aligned16 = ((uintptr_t)d_in % 16) == 0 && ((uintptr_t)d_out % 16) == 0 && in_col_stride % 2 == 0 &&
                           out_col_stride % 2 == 0;
    unsigned F = 0, Lr = 12;
    if (n_cosets <= 64 && front_policy) {
        if (front % 4 == 2) {          // 2^22: 4 + 8 + 10 measured 283.5 ms per proof, 5 + 8 + 9 284.5 (2 + 8 + 12: 284.7)
            if (front_policy < 5 && aligned16) F = 4, Lr = 10;
            else F = 5, Lr = 9;
        } else if (front % 4 == 3) {   // 2^23: 5 + 8 + 10 measured 560.0 ms per proof against 565.4 for 3 + 8 + 12
            F = 5, Lr = 10;
        }
    }
    if (F) {
        if (F == 5)
            launch_ntt_first5(src, d_out, d_tw, d_round_scale, log_n, n_cols, n_cosets, src_col_stride, src_coset_stride,
                              out_col_stride, s);
        else
            launch_ntt_first4(src, d_out, d_tw, d_round_scale, log_n, n_cols, n_cosets, src_col_stride, src_coset_stride,
                              out_col_stride, s);
        advance(F);
        front = log_n - F - Lr;
        while (front >= 8) {
            launch_ntt_strided8(src, d_out, d_tw, d_round_scale, log_n, r0, n_cols, n_cosets, src_col_stride,
                                src_coset_stride, out_col_stride, s);
            advance(8);
            front -= 8;
        }
        if (front == 4) {
            launch_ntt_strided4(src, d_out, d_tw, d_round_scale, log_n, r0, n_cols, n_cosets, src_col_stride,
                                src_coset_stride, out_col_stride, s);
            advance(4);
        }
        launch_ntt_local12(src, d_out, d_tw, d_round_scale, log_n, n_cols, n_cosets, src_col_stride, src_coset_stride,
                           out_col_stride, Lr, s);
        return;
    }
    if (front % 4) {
        unsigned R = front % 4;
        const bool old_remainder = bj::env().ntt_generic_remainder;
        if (!old_remainder && n_cosets <= 64)
            launch_first_rounds(src, d_out, d_tw, d_round_scale, log_n, R, n_cols, n_cosets, src_col_stride,
                                src_coset_stride, out_col_stride, s);
        else
            launch_generic_pass(src, d_out, d_tw, d_round_scale, log_n, r0, R, TILE_LOG - R, n_cols, n_cosets,
                                src_col_stride, src_coset_stride, out_col_stride, s);
        advance(R);
        front -= R;
    }
    while (front >= 8) {
        launch_ntt_strided8(src, d_out, d_tw, d_round_scale, log_n, r0, n_cols, n_cosets, src_col_stride,
                            src_coset_stride, out_col_stride, s);
        advance(8);
        front -= 8;
    }
    if (front == 4) {
        launch_ntt_strided4(src, d_out, d_tw, d_round_scale, log_n, r0, n_cols, n_cosets, src_col_stride,
                            src_coset_stride, out_col_stride, s);
        advance(4);
    }
    launch_ntt_local12(src, d_out, d_tw, d_round_scale, log_n, n_cols, n_cosets, src_col_stride, src_coset_stride,
                       out_col_stride, 12, s);
}

// ---------------------------------------------------------------------------------------------------------
// bit-reversal permutation (+ optional scaling by scale * step^i), out-of-place:  out[i] = in[bitrev(i)] * ...
// (bitreverse_enumeration_inplace fft/mod.rs:41-155; the n^-1 and coset^-i factors of ifft_natural_to_natural)
// ---------------------------------------------------------------------------------------------------------
__global__ void bitrev_scale_kernel(const u64 *in, u64 *out, unsigned log_n, size_t in_col_stride,
                                    size_t out_col_stride, u64 scale, u64 step) {
    size_t n = (size_t)1 << log_n;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 *src = in + (size_t)blockIdx.y * in_col_stride;
    u64 *dst = out + (size_t)blockIdx.y * out_col_stride;
    u64 v = gl::canon(src[gl::bitrev32((u32)i, log_n)]);
    if (scale != 1) v = gl::mul(v, scale);
    if (step != 1) v = gl::mul(v, gl::pow(step, i));
    dst[i] = v;
}
// Tiled variant for log_n >= 10: index i = [h : 5 bits][mid][l : 5 bits] and bitrev(i) = [rev(l)][rev(mid)][rev(h)], so for a
// fixed mid the 32 x 32 elements {(h, l)} are read as 32 runs of 32 consecutive words (row r = rev(l), column c = rev(h))
// and written as 32 runs of 32 consecutive words (row h, column l): both sides of the permutation are 256-byte runs, the
// transposition happens in LDS (row stride 33: conflict-free).  The gather above reads one word per 128-byte line.
// The factor scale * step^i of a coset transform splits along the same three index fields: scale * step^(h << (log_n - 5)) and
// step^l come from the host (64 words of kernel arguments), step^(32 mid) is one exponentiation per workgroup — two products
// per element where a per-element power cost ~45 (the quotient's 2 x 2^24 inverse transform: 695 -> ~90 us).
struct BitrevPowers {
    u64 hi[32];      // scale * step^(h << (log_n - 5))
    u64 lo[32];      // step^l
    u64 step32;      // step^32
};
__global__ void __launch_bounds__(256)
bitrev_scale_tiled_kernel(const u64 *in, u64 *out, unsigned log_n, size_t in_col_stride, size_t out_col_stride, u64 scale,
                          int stepped, BitrevPowers pw) {
    __shared__ u64 tile[32][33];
    __shared__ u64 pw_mid;
    const unsigned mid_bits = log_n - 10;
    const u32 mid = blockIdx.x, rmid = gl::bitrev32(mid, mid_bits);
    const u64 *src = in + (size_t)blockIdx.y * in_col_stride;
    u64 *dst = out + (size_t)blockIdx.y * out_col_stride;
    const unsigned c = threadIdx.x & 31, r0 = threadIdx.x >> 5;       // 8 rows per sweep
    if (stepped && threadIdx.x == 0) pw_mid = gl::pow(pw.step32, mid);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned r = r0 + 8 * k;
        tile[r][c] = src[((size_t)r << (log_n - 5)) | ((size_t)rmid << 5) | c];
    }
    __syncthreads();
    const u64 col_factor = stepped ? gl::mul(pw.lo[c], pw_mid) : 1;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned h = r0 + 8 * k, l = c;
        const size_t i = ((size_t)h << (log_n - 5)) | ((size_t)mid << 5) | l;
        u64 v = gl::canon(tile[gl::bitrev32(l, 5)][gl::bitrev32(h, 5)]);
        if (stepped)
            v = gl::mul(v, gl::mul(pw.hi[h], col_factor));
        else if (scale != 1)
            v = gl::mul(v, scale);
        dst[i] = v;
    }
}
void launch_bitrev_scale(const u64 *d_in, u64 *d_out, unsigned log_n, unsigned n_cols, size_t in_col_stride,
                         size_t out_col_stride, u64 scale, u64 step, hipStream_t s) {
    size_t n = (size_t)1 << log_n;
    unsigned tpb = 256;
    if (log_n >= 10 && !bj::env().bitrev_gather) {
        BitrevPowers pw{};
        const int stepped = step != 1;
        if (stepped) {
            const u64 sh = gl::pow(step, (u64)1 << (log_n - 5));
            u64 a = gl::canon(scale), b = 1;
            for (int k = 0; k < 32; k++) {
                pw.hi[k] = a;
                pw.lo[k] = b;
                a = gl::mul(a, sh);
                b = gl::mul(b, step);
            }
            pw.step32 = b;
        }
        hipLaunchKernelGGL(bitrev_scale_tiled_kernel, dim3(1u << (log_n - 10), n_cols), dim3(256), 0, s, d_in, d_out, log_n,
                           in_col_stride, out_col_stride, scale, stepped, pw);
    } else
        hipLaunchKernelGGL(bitrev_scale_kernel, dim3((unsigned)((n + tpb - 1) / tpb), n_cols), dim3(tpb), 0, s, d_in,
                           d_out, log_n, in_col_stride, out_col_stride, scale, step);
}

__global__ void canonicalize_kernel(u64 *a, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) a[i] = gl::canon(a[i]);
}
void launch_canonicalize(u64 *d, size_t n, hipStream_t s) {
    unsigned blocks = (unsigned)((n + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (blocks == 0) return;
    hipLaunchKernelGGL(canonicalize_kernel, dim3(blocks), dim3(256), 0, s, d, n);
}

// elementwise field operators on arbitrary u64 inputs (row a1/a2 of SURVEY §8 at operator level; field/goldilocks/mod.rs:188-255,
// field/traits/field.rs:407-512); op: 0 add, 1 sub, 2 mul, 3 mul through the weak (lazy) product of the hash kernels,
// 4 square, 5 inverse (0 -> 0), 9 / 10 weak sum / difference of the raw words, 11 weak F_p^2 product, 6 F_p^2 multiplication on (a0,a1) x (b0,b1) with the second halves at +n,
// 7 the NTT butterfly (u, v) <- (u + v*w, u - v*w) with a = [u | v], b = w, 8 the same with w = 1
__global__ void field_op_kernel(int op, const u64 *a, const u64 *b, u64 *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (op == 6) {
        gl::e2 x = {gl::canon(a[i]), gl::canon(a[n + i])}, y = {gl::canon(b[i]), gl::canon(b[n + i])};
        gl::e2 r = gl::e2_mul(x, y);
        out[i] = r.c0;
        out[n + i] = r.c1;
        return;
    }
    if (op == 11) {   // the same product on the RAW words (weak residues in, nothing canonicalised before the store)
        const gl::e2 r = gl::e2_mul_weak(gl::e2{a[i], a[n + i]}, gl::e2{b[i], b[n + i]});
        out[i] = gl::canon(r.c0);
        out[n + i] = gl::canon(r.c1);
        return;
    }
    if (op == 7 || op == 8) {   // two lazy butterflies per thread, exactly as the NTT kernels run them (n even)
        const size_t h = n / 2;
        if (i >= h) return;
        u64 ua = a[i], va = a[n + i], ub = a[i + h], vb = a[n + i + h];
        if (op == 7)
            gl::butterfly2_weak(ua, va, b[i], ub, vb, b[i + h]);
        else
            gl::addsub2_weak(ua, va, ub, vb);
        out[i] = gl::canon(ua); out[n + i] = gl::canon(va); out[i + h] = gl::canon(ub); out[n + i + h] = gl::canon(vb);
        return;
    }
    const u64 x = a[i], y = b ? b[i] : 0;
    u64 r;
    switch (op) {
    case 0: r = gl::add(gl::canon(x), gl::canon(y)); break;
    case 1: r = gl::sub(gl::canon(x), gl::canon(y)); break;
    case 2: r = gl::mul(x, y); break;
    case 3: r = gl::canon(gl::mul_weak(x, y)); break;
    case 4: r = gl::sqr(x); break;
    case 9: r = gl::canon(gl::add_weak(x, y)); break;
    case 10: r = gl::canon(gl::sub_weak(x, y)); break;
    default: r = gl::inv(gl::canon(x)); break;
    }
    out[i] = r;
}
void launch_field_op(int op, const u64 *a, const u64 *b, u64 *out, size_t n, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(field_op_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, op, a, b, out, n);
}

}  // namespace bj
