// HBM roofline microbenchmarks for the access shapes of the NTT front pass (ntt_first4: one read, eight writes per element, the
// writes spread over 16 x 8 streams 2^log_n / 16 elements apart) — what "achievable" means for a write-heavy kernel on this part.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_hbm.hip -o tools/microbench_hbm     Run: tools/microbench_hbm [log_n] [cols]
//   copy        out[i] = in[i]                       16-byte accesses, one read per write
//   fill        out[i] = f(i)                        writes only, contiguous
//   expand8     out[c][i] = in[i] + c, c < 8         one read, eight contiguous write streams        (9 words moved per input word)
//   expand8x16  the front pass's shape: lane = two adjacent i of a slice, 16 slices, 8 cosets: 16 loads, 128 16-byte stores,
//               no arithmetic beyond an add — the pass's traffic without its butterflies
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;

__global__ void __launch_bounds__(256) k_copy(const ulonglong2 *in, ulonglong2 *out, size_t n2) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n2) out[i] = in[i];
}
__global__ void __launch_bounds__(256) k_fill(ulonglong2 *out, size_t n2) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n2) out[i] = make_ulonglong2(i, ~i);
}
__global__ void __launch_bounds__(256) k_expand8(const ulonglong2 *in, ulonglong2 *out, size_t n2) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n2) return;
    ulonglong2 v = in[i];
#pragma unroll
    for (int c = 0; c < 8; c++) out[(size_t)c * n2 + i] = make_ulonglong2(v.x + c, v.y + c);
}
// per column: in[n], out[8][n]; lane owns i, i+1 of each of the 16 slices
__global__ void __launch_bounds__(256) k_expand8x16(const u64 *in, u64 *out, unsigned log_n) {
    const size_t n = (size_t)1 << log_n, sl = n >> 4;
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i >= sl) return;
    const u64 *src = in + (size_t)blockIdx.y * n + i;
    u64 *dst = out + (size_t)blockIdx.y * 8 * n + i;
    ulonglong2 x[16];
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = *reinterpret_cast<const ulonglong2 *>(src + (size_t)m * sl);
    for (unsigned c = 0; c < 8; c++) {
#pragma unroll
        for (int m = 0; m < 16; m++)
            *reinterpret_cast<ulonglong2 *>(dst + (size_t)c * n + (size_t)m * sl) = make_ulonglong2(x[m].x + c, x[m].y ^ c);
    }
}

int main(int argc, char **argv) {
    const unsigned log_n = argc > 1 ? atoi(argv[1]) : 22, cols = argc > 2 ? atoi(argv[2]) : 93;
    const size_t n = (size_t)1 << log_n, in_words = n * cols, out_words = 8 * in_words;
    u64 *in, *out;
    if (hipMalloc(&in, in_words * 8) != hipSuccess || hipMalloc(&out, out_words * 8) != hipSuccess) return 1;
    (void)hipMemset(in, 1, in_words * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    auto run = [&](const char *name, double bytes, auto launch) {
        launch();
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 5; r++) launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        ms /= 5;
        printf("{\"kernel\": \"%s\", \"ms\": %.3f, \"GBps\": %.1f}\n", name, ms, bytes / ms / 1e6);
    };
    const size_t big2 = out_words / 2;   // 16-byte elements of the big buffer
    run("copy (read n, write n) over half the big buffer", 8.0 * out_words, [&] {
        hipLaunchKernelGGL(k_copy, dim3((unsigned)((big2 / 2 + 255) / 256)), dim3(256), 0, 0, (const ulonglong2 *)out, (ulonglong2 *)out + big2 / 2, big2 / 2);
    });
    run("fill (write only)", 8.0 * out_words, [&] {
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((big2 + 255) / 256)), dim3(256), 0, 0, (ulonglong2 *)out, big2);
    });
    run("expand8 (1 read, 8 contiguous write streams)", 9.0 * 8 * in_words, [&] {
        hipLaunchKernelGGL(k_expand8, dim3((unsigned)((in_words / 2 + 255) / 256)), dim3(256), 0, 0, (const ulonglong2 *)in, (ulonglong2 *)out, in_words / 2);
    });
    run("expand8x16 (the front pass's access shape)", 9.0 * 8 * in_words, [&] {
        hipLaunchKernelGGL(k_expand8x16, dim3((unsigned)((n / 32 + 255) / 256), cols), dim3(256), 0, 0, in, out, log_n);
    });
    return 0;
}
