// Host harness for the fused gate sweep of csrc/gate_aot.hip (generated): the per-point logic of the fused kernel
// (gpaot::fused_point) against the per-gate logic (gpaot::aot_point) on random columns, for random subsets of the generated
// bodies, random selector paths, repetition strides and window sizes.  Both are __host__ __device__; the device kernels are thin
// wrappers around them.  Build + run (no GPU needed):
//   hipcc --cuda-host-only -x hip -std=c++17 -O1 -DBJ_GATE_AOT_HOST_ONLY -Iera_boojum_amd/csrc tools/gate_fused_host_check.cpp -o /tmp/gfhc && /tmp/gfhc
#include "../era_boojum_amd/csrc/gate_aot.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace bj;
using namespace bj::gpaot;

static uint64_t rng_state = 0x243F6A8885A308D3ULL;
static uint64_t rnd() {
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

int main() {
    const unsigned V = 130, NC = 12;
    const size_t Q = 48;
    std::vector<uint64_t> vars((size_t)V * Q), consts((size_t)NC * Q);
    int trials = 0;
    for (int trial = 0; trial < 400; trial++) {
        for (auto &x : vars) x = rnd();                    // any u64: operands are canonicalised on load
        for (auto &x : consts) x = (trial & 1) ? (rnd() & 1) : rnd();   // boolean selector columns half of the time
        const int n = 2 + (int)(rnd() % (BJ_FUSED_MAX - 1));
        FusedArgs f{};
        f.n = n;
        f.span = 0;
        f.window = 1 + (unsigned)(rnd() % 40);
        std::vector<std::vector<uint64_t>> alphas(n);
        std::vector<uint64_t> out0(Q), out1(Q), ref0, ref1;
        for (auto &x : out0) x = rnd();
        for (auto &x : out1) x = rnd();
        ref0 = out0;
        ref1 = out1;
        bool ok_geometry = true;
        for (int k = 0; k < n; k++) {
            int id;
            do id = (int)(rnd() % NUM_FUSABLE); while (BODY_INFO[id].wit_extent);   // the sweep takes the light bodies only
            const BodyInfo &B = BODY_INFO[id];
            ProgArgs a{};
            a.vars = vars.data(); a.var_stride = Q; a.consts = consts.data(); a.const_stride = Q;
            a.path_len = (unsigned)(rnd() % 4);
            for (unsigned b = 0; b < a.path_len; b++) a.path[b] = (unsigned char)(rnd() & 1);
            a.rep_var_stride = (unsigned)B.var_extent + (unsigned)(rnd() % 3);
            const unsigned max_reps = (V - (unsigned)B.var_extent) / a.rep_var_stride + 1;
            a.reps = 1 + (unsigned)(rnd() % max_reps);
            const bool per_rep_consts = (rnd() & 1) && B.const_extent && a.path_len + a.reps * (unsigned)B.const_extent <= NC;
            a.rep_const_stride = per_rep_consts ? (unsigned)B.const_extent : 0;
            if (a.path_len + (unsigned)B.const_extent > NC) ok_geometry = false;
            alphas[k].resize((size_t)2 * a.reps * B.num_terms);
            for (auto &x : alphas[k]) x = rnd() % gl::P;
            a.alphas = alphas[k].data();
            a.Q = Q; a.out0 = out0.data(); a.out1 = out1.data(); a.terms = nullptr; a.wits = nullptr;
            a.n_writes = (unsigned)B.num_terms;
            f.g[k] = a;
            f.id[k] = id;
            const unsigned cover = (a.reps - 1) * a.rep_var_stride + 1;
            if (cover > f.span) f.span = cover;
        }
        if (!ok_geometry) continue;
        trials++;
        // reference: gate after gate, each adding sel * sum alpha * term
        for (int k = 0; k < n; k++) {
            ProgArgs a = f.g[k];
            a.out0 = ref0.data();
            a.out1 = ref1.data();
            for (size_t I = 0; I < Q; I++) aot_point_by_id(f.id[k], a, I);
        }
        for (size_t I = 0; I < Q; I++) {
            uint64_t sel[BJ_FUSED_MAX];
            fused_point(f, I, sel, 1);
        }
        for (size_t I = 0; I < Q; I++)
            if (out0[I] != ref0[I] || out1[I] != ref1[I]) {
                std::printf("MISMATCH trial %d point %zu: fused (%016llx, %016llx) reference (%016llx, %016llx), %d gates, window %u, span %u\n",
                            trial, I, (unsigned long long)out0[I], (unsigned long long)out1[I], (unsigned long long)ref0[I],
                            (unsigned long long)ref1[I], n, f.window, f.span);
                return 1;
            }
    }
    std::printf("fused == per-gate on %d random gate sets (%zu points each)\n", trials, Q);
    return trials > 100 ? 0 : 2;
}
