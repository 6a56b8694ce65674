// Internal: host-side declarations around the device runtime of the op-list gates (gate_body_rt.h): the generated kernels of
// gate_aot.hip (selected by the structural fingerprint of gate_canon.h) and the run-time compiled ones of gate_jit.hip.
#pragma once
#include "gate_program.h"
#include "gate_body_rt.h"

namespace bj {
// true if every program has a generated body and the launch was made (quotient mode only: alphas, out0 / out1 of args[0])
bool launch_gate_aot_fused(const uint64_t *fp0, const uint64_t *fp1, const gpdev::ProgArgs *args, unsigned n, unsigned blocks,
                           hipStream_t s);
// true if a generated kernel exists for the program with this fingerprint (and it was launched)
bool launch_gate_aot(uint64_t fp0, uint64_t fp1, const gpdev::ProgArgs &a, unsigned blocks, hipStream_t s);
bool gate_aot_known(uint64_t fp0, uint64_t fp1);
bool gate_aot_fusable(uint64_t fp0, uint64_t fp1);   // known AND light enough for the fused sweep
// the fingerprint of the reference's own capture of Poseidon2FlattenedGate<8,12,4> without witness columns
// (src/cs/gates/poseidon2.rs:166-391): such a program is run by the hand-written evaluator of gate_poseidon2.hip
bool gate_is_poseidon2_flattened(uint64_t fp0, uint64_t fp1);
// a fingerprint hit is believed only with the structural summary recorded at build time for that body (gate_aot.hip)
bool gate_aot_summary_matches(uint64_t fp0, uint32_t num_ops, uint32_t num_slots, uint32_t num_terms, uint32_t var_extent,
                              uint32_t const_extent, uint32_t wit_extent);
}  // namespace bj
