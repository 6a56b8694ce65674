// Internal: run-time compilation of a host's own gate op lists (seam S3).  SURVEY §8b asks for a "JIT/AOT-generated device
// function" per captured evaluator: the evaluators the reference ships have build-time kernels (gate_aot.hip); any other
// GPUDataCapture — a new gate, a parametrised one (MatrixMultiplicationGate with the host's matrix) — is compiled once per
// process with hiprtc from the canonical program (gate_canon.h: the same body text the build-time generator emits) and cached
// by its structural fingerprint.  hiprtc is dlopen'ed; without it, or with BJ_GATE_NO_JIT set, the program runs in the
// interpreter of gate_program.hip — a slower kernel, same results.
#pragma once
#include "gate_canon.h"
#include "gate_program_dev.h"

namespace bj {
struct JitKernel;
// nullptr when the compiler is unavailable or the compilation failed (the reason is kept for bj_gate_jit_status)
const JitKernel *jit_gate_kernel(bj_ctx *ctx, const canon::Program &C);
bool launch_jit_gate(const JitKernel *k, const gpdev::ProgArgs &a, unsigned blocks, hipStream_t s);
// the HIP source jit_gate_kernel compiles for this program (also what bj_gate_program_jit_source returns)
std::string jit_gate_source(const canon::Program &C);
}  // namespace bj
