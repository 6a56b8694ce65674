// Integer-ALU roofline microbenchmarks for gfx950 (BASELINE.md §5 "measure first").
// Build: hipcc --offload-arch=gfx950 -O3 -I era_boojum_amd/csrc tools/microbench.hip -o tools/microbench
// Prints wave-instruction throughput of the building blocks of a Goldilocks multiplication and the resulting
// field-mul / butterfly / Poseidon2-sbox rates for the whole chip.
#include "gl.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

using gl::u64;
using gl::u32;

#define ITERS 4096
#define CHAINS 8

template <int OP>
__global__ void __launch_bounds__(256) ubench(u64 *out, u64 seed) {
    u64 x[CHAINS];
    u64 w = seed | 1;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = seed + threadIdx.x * 977 + c * 131 + blockIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            if (OP == 0) {  // v_mad_u64_u32
                x[c] = (u64)(u32)x[c] * (u32)w + x[c];
            } else if (OP == 1) {  // v_mul_lo_u32
                x[c] = (u32)((u32)x[c] * (u32)w) + 1;
            } else if (OP == 2) {  // v_mul_hi_u32
                x[c] = __umulhi((u32)x[c], (u32)w) + (u32)x[c];
            } else if (OP == 3) {  // 64-bit add (2 x v_add)
                x[c] = x[c] + w;
            } else if (OP == 4) {  // gl::mul
                x[c] = gl::mul(x[c], w);
            } else if (OP == 5) {  // gl::add
                x[c] = gl::add(x[c], w);
            } else if (OP == 6) {  // gl::sub
                x[c] = gl::sub(x[c], w);
            } else if (OP == 7) {  // butterfly: (u, v*w) -> (u+vw, u-vw) on chain pairs
                if ((c & 1) == 0) {
                    u64 v = gl::mul(x[c + 1], w);
                    u64 u = x[c];
                    x[c] = gl::add(u, v);
                    x[c + 1] = gl::sub(u, v);
                }
            } else if (OP == 8) {  // x^7
                u64 x2 = gl::sqr(x[c]), x3 = gl::mul(x2, x[c]), x4 = gl::sqr(x2);
                x[c] = gl::mul(x4, x3);
            } else if (OP == 9) {  // mul_pow2 (shift multiply)
                x[c] = gl::mul_pow2(x[c], 13);
            } else if (OP == 10) { // v_mul_u32_u24 (full rate?)
                x[c] = (u64)__umul24((u32)x[c], (u32)w) + 1;
            }
        }
    }
    u64 acc = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) acc ^= x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int OP>
double run(const char *name, double ops_per_iter_chain, u64 *d_out) {
    int blocks = 256 * 8, tpb = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(ubench<OP>, dim3(blocks), dim3(tpb), 0, 0, d_out, 0x9E3779B97F4A7C15ULL);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(ubench<OP>, dim3(blocks), dim3(tpb), 0, 0, d_out, 0x9E3779B97F4A7C15ULL);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double total = 3.0 * blocks * tpb * (double)ITERS * CHAINS * ops_per_iter_chain;
    double rate = total / (ms * 1e-3);
    // cycles per wave-op per SIMD at 2.4 GHz, 1024 SIMDs
    double cyc = 1024.0 * 2.4e9 / (rate / 64.0);
    printf("%-28s %10.3f Gop/s   (~%6.1f cycles per wave-op per SIMD @2.4GHz)\n", name, rate / 1e9, cyc);
    return rate;
}

int main() {
    u64 *d_out;
    hipMalloc(&d_out, 256 * 8 * 256 * sizeof(u64));
    run<0>("v_mad_u64_u32", 1, d_out);
    run<1>("v_mul_lo_u32(+add)", 1, d_out);
    run<2>("v_mul_hi_u32(+add)", 1, d_out);
    run<10>("v_mul_u32_u24(+add)", 1, d_out);
    run<3>("add u64", 1, d_out);
    run<5>("gl::add", 1, d_out);
    run<6>("gl::sub", 1, d_out);
    run<9>("gl::mul_pow2", 1, d_out);
    run<4>("gl::mul", 1, d_out);
    run<7>("butterfly (mul+add+sub)", 0.5, d_out);
    run<8>("x^7 (4 mul)", 1, d_out);
    hipFree(d_out);
    return 0;
}
