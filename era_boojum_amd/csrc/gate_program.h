// Internal: device-side form of a bj_gate_program (seam S3), shared by gate_program.hip and prover.hip.
#pragma once
#include "ctx.h"
#include "gate_body_rt.h"

// slots of the interpreter's private (scratch-memory) variant; the canonical form of gate_canon.h needs far fewer for every
// evaluator of the reference (13 for a 12 x 12 matrix gate, ~40 for the Poseidon2 flattened capture); a program whose
// schedule needs more than BJ_GATE_PROGRAM_MAX_SLOTS live values at once is refused
#define BJ_GATE_PROGRAM_MAX_TEMPORARIES 160
#define BJ_GATE_PROGRAM_MAX_SLOTS 1024

namespace bj {
struct JitKernel;             // gate_jit.hip: a kernel compiled at run time from the canonical program
struct DevProgram {
    void *block = nullptr;    // one allocation: values | relations
    gl::u64 *d_values = nullptr;
    DevRelation *d_rel = nullptr;
    unsigned n_rel = 0, n_writes = 0, n_tmp = 0;   // n_rel counts the OP_WRITE pseudo relations; n_tmp: slots
    bool reads_witness = false;   // the program has BJ_IDX_WITNESS_POLY operands
    unsigned var_extent = 0, const_extent = 0, wit_extent = 0;
    uint64_t fp[2] = {0, 0};      // structural fingerprint (gate_canon.h): selects a generated kernel when one exists
    const JitKernel *jit = nullptr;   // compiled at upload when no generated kernel exists (owned by the process-wide cache)
    int upload(bj_ctx *ctx, const bj_gate_program *p);   // canonicalises, packs and copies the program
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
