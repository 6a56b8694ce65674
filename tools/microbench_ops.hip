// Per-instruction issue cost of the integer VALU ops used by the Goldilocks arithmetic, measured with inline asm so
// the compiler cannot fold anything.  Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_ops.hip -o tools/microbench_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 2048
#define REP8(x) x x x x x x x x
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out) {
    uint32_t a = threadIdx.x, b = blockIdx.x + 3, c = 7, d = 11, e = 13, f = 17, g = 19, h = 23;
    uint64_t A = a, B = b, C = c, D = d;
    uint64_t mask = 0x5555555555555555ull + blockIdx.x;
    if (OP == 28) asm volatile("s_mov_b64 vcc, %0" : : "s"(mask) : "vcc");
    if (OP == 29) asm volatile("s_mov_b64 s[10:11], %0\n s_mov_b64 s[12:13], %0\n s_mov_b64 s[14:15], %0\n s_mov_b64 s[16:17], %0" : : "s"(mask) : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17");
    for (int it = 0; it < ITERS; it++) {
        if (OP == 0) { REP8(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : );) }
        if (OP == 1) { REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_add_co_u32 %2, vcc, %2, %1\n v_add_co_u32 %3, vcc, %3, %1\n v_add_co_u32 %4, vcc, %4, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");) }
        if (OP == 2) { REP8(asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc\n v_addc_co_u32 %2, vcc, %2, %1, vcc\n v_addc_co_u32 %3, vcc, %3, %1, vcc\n v_addc_co_u32 %4, vcc, %4, %1, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");) }
        if (OP == 3) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %1, vcc\n v_cndmask_b32 %3, %3, %1, vcc\n v_cndmask_b32 %4, %4, %1, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");) }
        if (OP == 4) { REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %2, %2, 0, %1\n v_lshl_add_u64 %3, %3, 0, %1\n v_lshl_add_u64 %4, %4, 0, %1" : "+v"(A), "+v"(B), "+v"(C), "+v"(D) : : );) }
        if (OP == 5) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(A), "+v"(B), "+v"(C), "+v"(D) : "v"(e), "v"(f) : "vcc");) }
        if (OP == 6) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %1\n v_mul_lo_u32 %3, %3, %1\n v_mul_lo_u32 %4, %4, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : );) }
        if (OP == 7) { REP8(asm volatile("v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %2, %2, %1\n v_mul_hi_u32 %3, %3, %1\n v_mul_hi_u32 %4, %4, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : );) }
        if (OP == 8) { REP8(asm volatile("v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %2, %2, %1, %3\n v_mad_u32_u24 %3, %3, %1, %4\n v_mad_u32_u24 %4, %4, %1, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : );) }
        if (OP == 9) { REP8(asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_cmp_lt_u64 vcc, %1, %2\n v_cmp_lt_u64 vcc, %2, %3\n v_cmp_lt_u64 vcc, %3, %0" : : "v"(A), "v"(B), "v"(C), "v"(D) : "vcc");) }
        if (OP == 10) { REP8(asm volatile("v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %2, %2, %1, %3\n v_add3_u32 %3, %3, %1, %4\n v_add3_u32 %4, %4, %1, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : );) }
        if (OP == 11) { REP8(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %2, %2, %1, %3\n v_fma_f32 %3, %3, %1, %4\n v_fma_f32 %4, %4, %1, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : );) }
        if (OP == 12) { REP8(asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %2, %2, %1, %3\n v_fma_f64 %3, %3, %1, %0\n v_fma_f64 %1, %1, %2, %3" : "+v"(A), "+v"(B), "+v"(C), "+v"(D) : : );) }
        if (OP == 13) { REP8(asm volatile("v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %2, %2, %1\n v_mul_u32_u24 %3, %3, %1\n v_mul_u32_u24 %4, %4, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : );) }
        if (OP == 14) { REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_add_u32 %1, %1, %4\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "vcc");) }
        if (OP == 15) { REP8(asm volatile("v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %2, %2, %1\n v_pk_add_u16 %3, %3, %1\n v_pk_add_u16 %4, %4, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : );) }
        if (OP == 16) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_add_u32 %2, %2, %4\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_add_u32 %3, %3, %4" : "+v"(A), "+v"(B), "+v"(a), "+v"(b) : "v"(e), "v"(f) : "vcc");) }
        if (OP == 17) { REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_add_u32 %2, %2, %4\n v_lshl_add_u64 %1, %1, 0, %0\n v_add_u32 %3, %3, %4" : "+v"(A), "+v"(B), "+v"(a), "+v"(b) : "v"(e) : );) }
        if (OP == 18) { REP8(asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_add_u32 %2, %2, %4\n v_cmp_lt_u64 vcc, %1, %0\n v_add_u32 %3, %3, %4" : : "v"(A), "v"(B), "v"(a), "v"(b), "v"(e) : "vcc");) }
        if (OP == 19) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, %4\n v_cndmask_b32 %2, %2, %1, %4\n v_cndmask_b32 %3, %3, %1, %4\n v_cndmask_b32 %1, %1, %0, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(mask) : );) }
        if (OP == 20) { REP8(asm volatile("v_alignbit_b32 %0, %0, %1, 7\n v_alignbit_b32 %2, %2, %1, 7\n v_alignbit_b32 %3, %3, %1, 7\n v_alignbit_b32 %1, %1, %0, 7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : );) }
        if (OP == 21) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_add_co_u32 %2, vcc, %2, %4\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_add_co_u32 %3, vcc, %3, %4" : "+v"(A), "+v"(B), "+v"(a), "+v"(b) : "v"(e), "v"(f) : "vcc");) }
        if (OP == 22) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4" : "+v"(A), "+v"(b), "+v"(a), "+v"(c) : "v"(e), "v"(f) : "vcc");) }
        if (OP == 23) { REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %4" : "+v"(d), "+v"(b), "+v"(a), "+v"(c) : "v"(e), "v"(f) : "vcc");) }
        if (OP == 24) { REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %0\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(A), "+v"(B), "+v"(C), "+v"(D) : "v"(e), "v"(f) : "vcc");) }
        if (OP == 26) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, %4\n v_add_u32 %2, %2, %1\n v_cndmask_b32 %3, %3, %1, %4\n v_add_u32 %1, %1, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(mask) : );) }
        if (OP == 27) { REP8(asm volatile("v_cmp_lt_u64 s[10:11], %4, %5\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_cndmask_b32 %0, %0, %1, s[10:11]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(A), "v"(B) : "s10", "s11");) }
        if (OP == 28) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %1, vcc\n v_cndmask_b32 %3, %3, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : );) }
        if (OP == 29) { REP8(asm volatile("v_subb_co_u32 %0, s[10:11], %0, %1, s[10:11]\n v_subb_co_u32 %2, s[12:13], %2, %1, s[12:13]\n v_subb_co_u32 %3, s[14:15], %3, %1, s[14:15]\n v_subb_co_u32 %1, s[16:17], %1, %0, s[16:17]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17");) }
        if (OP == 30) { REP8(asm volatile("v_add_co_u32 %0, s[10:11], %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_addc_co_u32 %1, s[10:11], %1, %0, s[10:11]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "s10", "s11");) }
        if (OP == 31) { REP8(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");) }
        if (OP == 32) { REP8(asm volatile("v_cmp_lt_u32 s[10:11], %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_cndmask_b32 %0, %0, %1, s[10:11]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "s10", "s11");) }
        if (OP == 33) { REP8(asm volatile("v_add_co_u32 %1, vcc, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");) }
        if (OP == 34) { REP8(asm volatile("v_add_co_u32 %1, s[10:11], %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_cndmask_b32 %0, %0, %1, s[10:11]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "s10", "s11");) }
        if (OP == 25) { REP8(asm volatile("v_xor_b32 %0, %0, %1\n v_and_b32 %2, %2, %1\n v_lshlrev_b32 %3, 3, %3\n v_sub_u32 %1, %1, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : );) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ (uint32_t)(A ^ B ^ C ^ D) ^ g ^ h;
}
template <int OP> void run(const char *name, uint32_t *d) {
    printf("[%d] ", OP); fflush(stdout);
    int blocks = 256 * 8;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double insts = 3.0 * blocks * 4 /*waves per block*/ * (double)ITERS * 32;
    double per_simd = insts / 1024.0;
    printf("%-34s %7.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, ms * 1e-3 * 2.4e9 / per_simd);
}
int main() {
    uint32_t *d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_add_u32", d); run<1>("v_add_co_u32", d); run<2>("v_addc_co_u32 (dependent on vcc)", d);
    run<14>("add_co/add/addc/add interleaved", d);
    run<3>("v_cndmask_b32", d); run<10>("v_add3_u32", d); run<15>("v_pk_add_u16", d);
    run<4>("v_lshl_add_u64", d); run<9>("v_cmp_lt_u64", d);
    run<5>("v_mad_u64_u32", d); run<6>("v_mul_lo_u32", d); run<7>("v_mul_hi_u32", d); run<8>("v_mad_u32_u24", d); run<13>("v_mul_u32_u24", d);
    run<16>("mad_u64 + add_u32 interleaved", d); run<22>("1 mad_u64 + 3 add_u32", d); run<23>("1 add_co + 3 add_u32", d);
    run<17>("lshl_add_u64 + add_u32 interleaved", d);
    run<19>("v_cndmask_b32 e64 (sgpr mask)", d); run<20>("v_alignbit_b32", d); run<21>("mad_u64 + add_co interleaved", d);
    run<24>("2 lshl_add_u64 + 2 mad_u64", d); run<25>("xor/and/lshl/sub plain mix", d);
    run<26>("cndmask e64 + add_u32 interleaved", d); run<27>("cmp_lt_u64 -> cndmask e64 (dependent pair) x2", d); run<28>("v_cndmask_b32 e32 (vcc set once)", d);
    run<29>("v_subb_co_u32 e64 (sgpr pair carry)", d); run<30>("add_co e64 -> addc_co e64 (sgpr pair) + 2 plain", d);
    run<31>("cmp_lt_u32 vcc -> cndmask e32 vcc + 2 plain", d); run<32>("cmp_lt_u32 sgpr -> cndmask e64 + 2 plain", d);
    run<33>("add_co vcc -> cndmask e32 vcc + 2 plain", d); run<34>("add_co sgpr -> cndmask e64 + 2 plain", d);
    run<11>("v_fma_f32", d); run<12>("v_fma_f64", d);
    return 0;
}
