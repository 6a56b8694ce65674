// Internal: device-side form of a bj_gate_program (seam S3), shared by gate_program.hip and prover.hip.
#pragma once
#include "ctx.h"

#define BJ_GATE_PROGRAM_MAX_TEMPORARIES 160

namespace bj {
struct DevRelation {
    uint32_t op, dst, a, b;   // a, b: kind << 28 | index
};
struct DevProgram {
    void *block = nullptr;    // one allocation: values | relations | writes
    gl::u64 *d_values = nullptr;
    DevRelation *d_rel = nullptr;
    uint32_t *d_writes = nullptr;
    unsigned n_rel = 0, n_writes = 0, n_tmp = 0;
    bool reads_witness = false;   // the program has BJ_IDX_WITNESS_POLY operands
    uint64_t hash = 0, check = 0;   // two fingerprints of the program's content: select a generated kernel when one exists
    int upload(bj_ctx *ctx, const bj_gate_program *p);   // validates, packs and copies the program
    void release();
};
// highest variable / constant column index (relative to the repetition) the program reads, + 1; 0 if it reads none
void gate_program_extent(const bj_gate_program *p, unsigned *var_extent, unsigned *const_extent, unsigned *wit_extent = nullptr);
// quotient mode (d_alphas != nullptr): out += selector * sum alpha * term; stand-alone mode (d_terms != nullptr): raw terms
void launch_gate_program(const DevProgram &P, const gl::u64 *d_vars, size_t var_stride, const gl::u64 *d_consts,
                         size_t const_stride, unsigned path_len, const unsigned char *path, unsigned reps,
                         unsigned rep_var_stride, unsigned rep_const_stride, const gl::u64 *d_alphas, size_t Q,
                         gl::u64 *d_out0, gl::u64 *d_out1, gl::u64 *d_terms, hipStream_t s, const gl::u64 *d_wits = nullptr,
                         unsigned rep_wit_stride = 0);
// one op-list gate of a circuit for launch_gate_programs (quotient mode)
struct GateLaunch {
    const DevProgram *program;
    unsigned path_len;
    unsigned char path[8];
    unsigned reps, rep_var_stride, rep_const_stride, rep_wit_stride;
    const gl::u64 *d_alphas;
};
void launch_gate_programs(const GateLaunch *gates, unsigned n, const gl::u64 *d_vars, size_t var_stride, const gl::u64 *d_consts,
                          size_t const_stride, size_t Q, gl::u64 *d_out0, gl::u64 *d_out1, hipStream_t s, const gl::u64 *d_wits);
}  // namespace bj
