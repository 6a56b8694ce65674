// Internal: canonical form of a bj_gate_program (seam S3, include/boojum_hip.h).
//
// The reference's GPUDataCapture::from_evaluator (src/gpu_synthesizer/mod.rs:354-444) records one relation per arithmetic call
// of GpuSynthesizerFieldLike, each into a FRESH temporary taken from a process-wide counter (:210-352): what a Rust host hands
// over is an SSA list whose numbering depends on what was captured before it and whose order is the evaluator's call order.
// Everything the library does with such a list — the interpreter's slot budget, the choice of a generated kernel, the text of
// a run-time compiled one — starts from the form computed here, which depends on neither:
//   * the list becomes a DAG (hash-consed: equal sub-expressions are one node; commutative operands ordered by structural
//     hash; x+0, x-0, x*1, x*0, x+x, x*x, 0-x and operations on two constants are rewritten; dead relations dropped);
//   * a 128-bit structural fingerprint of the terms (Merkle hashes of the DAG) identifies the evaluator's FUNCTION: it is
//     invariant under renumbering of temporaries, reordering of independent relations and the peepholes above;
//   * a schedule (depth-first from the terms, the operand with the larger Sethi-Ullman number first, each term emitted as
//     soon as its value exists) and a linear-scan slot allocation over it: the 72 relations of U8x4FMAGate need 3 slots, a
//     288-relation matrix gate 2 to 25 (the dense external Poseidon2 matrix shares its products across rows), the
//     ~9.6 k-relation Poseidon2 flattened capture 42.
// Pure C++ (no HIP): built into libboojum_hip.so and, by build.py, into a host-only helper that gate_codegen.py uses to
// emit csrc/gate_aot.hip — one implementation, so the fingerprints of the generated kernels are the library's by construction.
#pragma once
#include "../../include/boojum_hip.h"

#include <cstdint>
#include <string>
#include <vector>

namespace bj {
namespace canon {

constexpr uint32_t OP_WRITE = 8;   // internal pseudo relation: operand `a` is quotient term `dst` of the repetition

struct Operand {
    uint32_t kind;    // bj_index_kind; BJ_IDX_TEMPORARY: a node of Program::nodes, BJ_IDX_CONSTANT_VALUE: an index into values
    uint32_t index;
};
struct Node {
    uint32_t op;      // bj_gate_op or OP_WRITE
    uint32_t dst;     // slot written (term number for OP_WRITE)
    Operand a, b;     // temporaries are named by NODE number here; slot_of[node] gives the slot
};
struct Program {
    std::vector<Node> nodes;            // in schedule order; operands of kind TEMPORARY refer to earlier entries
    std::vector<uint32_t> slot_of;      // per node (unused for OP_WRITE)
    std::vector<uint64_t> values;       // canonical residues, deduplicated, in first-use order
    uint32_t num_terms = 0, num_slots = 0, num_ops = 0;   // num_ops: nodes that are not OP_WRITE
    uint64_t fp[2] = {0, 0};            // structural fingerprint
    uint32_t var_extent = 0, const_extent = 0, wit_extent = 0;   // highest column index read + 1, per kind
};

// BJ_OK, or BJ_ERR_INVALID_ARG with a message (bad operand, use of an undefined temporary, ...)
int canonicalize(const bj_gate_program *p, Program *out, std::string *err);

// straight-line C++ statements computing the terms of one repetition: `const u64 n<i> = gl::op(...);` per node and
// `term[t] = ...;` per write, over the VAR(k) / CON(k) / WIT(k) macros of gate_body_rt.h
std::string emit_body(const Program &P, const char *indent);

}  // namespace canon
}  // namespace bj

// host-only C entry points of the helper library (also exported by libboojum_hip.so); no GPU needed
extern "C" {
// fingerprint + sizes of the canonical form; any of the out pointers may be NULL
int bj_gate_program_canonical_info(const bj_gate_program *program, uint64_t fp[2], uint32_t *num_slots, uint32_t *num_ops,
                                   uint32_t *extents3);
// body text (NUL-terminated) into `out` (capacity `cap`); returns the length needed including the terminator, 0 on error
size_t bj_gate_program_emit_body(const bj_gate_program *program, char *out, size_t cap);
}
