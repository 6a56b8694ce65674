"""What `GPUDataCapture::from_evaluator` records (src/gpu_synthesizer/mod.rs:210-352, 385-444), emulated for every evaluator of
src/cs/gates/: every arithmetic call of `GpuSynthesizerFieldLike` takes a FRESH temporary from a process-wide counter (so a
capture's temporaries start wherever the previous captures stopped and are never reused), relations come in the order
`evaluate_once` issues its calls, `mul_and_accumulate_into` / `small_pow` are the trait's defaults (field_like.rs:71-106).

Three evaluators are transliterated here a second time, independently of the product (`capture_fma`, `capture_zero_check`,
`capture_uintx_add`); the others run the `evaluate_*` bodies of era_boojum_amd/gate_program.py — each written call by call
from the reference's `evaluate_once` — under this module's recording context, i.e. with the reference's numbering.  What
checks those bodies as FUNCTIONS are independent formulas (oracle/gates.py, pinned by the reference's own proof, and
tests/test_gate_programs.py).

Round-4 audit: the `evaluate_*` bodies of era_boojum_amd/gate_program.py (and the trait defaults `mul_and_accumulate_into` /
`small_pow`, field_like.rs:70-106) were read again side by side with the reference's `evaluate_once` for reduction, constants
allocator, boolean, selection, parallel selection, conditional swap, dot product, quadratic combination, reduction by powers,
simple non-linearity, u32 add, u32 sub, FMA in the extension and the matrix gate: same calls in the same order, statement by
statement.  A reference-PRODUCED capture would still be the only pin of the order; no Rust toolchain here can produce one.

`to_program` is the Python mirror of `OwnedProgram::from_capture` in rust/prove_hip.rs (dense renumbering in definition order,
field constants into the value table); `to_program_raw` keeps the reference's sparse numbers."""
from era_boojum_amd import gate_program as GP

P = GP.P
_counter = [977]            # as if other gates had been captured before in this process


class V:
    def __init__(self, idx):
        self.idx = idx      # ("var" | "wit" | "const" | "tmp" | "value", number)


_OPNAME = {GP.OP_ADD: "Add", GP.OP_SUB: "Sub", GP.OP_MUL: "Mul", GP.OP_DOUBLE: "Double", GP.OP_NEGATE: "Negate",
           GP.OP_SQUARE: "Square", GP.OP_INVERSE: "Inverse"}


class Capture(GP.EvaluatorContext):
    def __init__(self):
        self.relations, self.writes = [], []

    def _fresh(self, rel):
        _counter[0] += 1
        t = V(("tmp", _counter[0]))
        self.relations.append((t.idx, rel))
        return t

    def _emit(self, op, a, b=None):          # EvaluatorContext's hook: one relation, one fresh temporary
        return self._fresh((_OPNAME[op], a.idx) if b is None else (_OPNAME[op], a.idx, b.idx))
    # the same calls spelled out, for the independent transliterations below
    def add(self, x, y): return self._fresh(("Add", x.idx, y.idx))
    def sub(self, x, y): return self._fresh(("Sub", x.idx, y.idx))
    def mul(self, x, y): return self._fresh(("Mul", x.idx, y.idx))
    def var(self, i): return V(("var", i))
    def wit(self, i): return V(("wit", i))
    def const_poly(self, i): return V(("const", i))
    const = const_poly
    def value(self, x): return V(("value", x % P))
    def push(self, x): self.writes.append(x.idx)


def capture_fma():          # fma_gate_without_constant.rs:96-126
    c = Capture()
    a, b, cc, d = (c.var(i) for i in range(4))
    q, l = c.const(0), c.const(1)
    contribution = c.mul(cc, l)
    t = c.mul(a, b)
    contribution = c.mul_and_accumulate_into(contribution, q, t)
    contribution = c.sub(contribution, d)
    c.push(contribution)
    return c


def capture_zero_check(use_witness_column_for_inversion=False):      # zero_check.rs:143-175
    c = Capture()
    one = c.value(1)
    inp, flag = c.var(0), c.var(1)
    inv = c.wit(0) if use_witness_column_for_inversion else c.var(2)
    contribution = c.mul_and_accumulate_into(flag, inp, inv)
    contribution = c.sub(contribution, one)
    c.push(contribution)
    c.push(c.mul(inp, flag))
    return c


def capture_uintx_add():    # uintx_add.rs:101-140
    c = Capture()
    shift = c.const(0)
    a, b, carry_in, cc, carry_out = (c.var(i) for i in range(5))
    contribution = c.add(a, b)
    contribution = c.add(contribution, carry_in)
    contribution = c.sub(contribution, cc)
    tmp = c.mul(shift, carry_out)
    contribution = c.sub(contribution, tmp)
    c.push(contribution)
    contribution = c.mul(carry_out, carry_out)
    contribution = c.sub(contribution, carry_out)
    c.push(contribution)
    return c


def capture(evaluate, *args, **kw):
    """An `evaluate_*` body of gate_program.py under the reference's recording context."""
    c = Capture()
    evaluate(c, *args, **kw)
    return c


MATRIX = [[(7 * r + 3 * k + 1) * 65537 % 99991 + 1 for k in range(12)] for r in range(12)]     # a host's own 12 x 12 matrix


def all_captures():
    """name -> (capture thunk, variables, constants, witness columns one repetition reads): every evaluator the reference
    compiles (src/cs/gates/mod.rs:117-143; nop / public input / lookup markers have no terms, the bounded wrappers run their
    inner evaluator)."""
    E = GP
    return {
        "fma": (capture_fma, 4, 2, 0),
        "zero_check": (capture_zero_check, 3, 0, 0),
        "zero_check_witness_inversion": (lambda: capture_zero_check(True), 2, 0, 1),
        "uintx_add": (capture_uintx_add, 5, 1, 0),
        "fma_product_body": (lambda: capture(E.evaluate_fma), 4, 2, 0),
        "reduction4": (lambda: capture(E.evaluate_reduction, 4), 5, 4, 0),
        "constants_allocator": (lambda: capture(E.evaluate_constants_allocator), 1, 1, 0),
        "boolean": (lambda: capture(E.evaluate_boolean), 1, 0, 0),
        "selection": (lambda: capture(E.evaluate_selection), 4, 0, 0),
        "parallel_selection4": (lambda: capture(E.evaluate_parallel_selection, 4), 13, 0, 0),
        "conditional_swap1": (lambda: capture(E.evaluate_conditional_swap, 1), 5, 0, 0),
        "conditional_swap2": (lambda: capture(E.evaluate_conditional_swap, 2), 9, 0, 0),
        "dot_product4": (lambda: capture(E.evaluate_dot_product, 4), 9, 0, 0),
        "quadratic_combination4": (lambda: capture(E.evaluate_quadratic_combination, 4), 8, 0, 0),
        "reduction_by_powers4": (lambda: capture(E.evaluate_reduction_by_powers, 4), 5, 1, 0),
        "simple_non_linearity7": (lambda: capture(E.evaluate_simple_non_linearity, 7), 2, 1, 0),
        "simple_non_linearity5": (lambda: capture(E.evaluate_simple_non_linearity, 5), 2, 1, 0),
        "simple_non_linearity3": (lambda: capture(E.evaluate_simple_non_linearity, 3), 2, 1, 0),
        "u32_add": (lambda: capture(E.evaluate_u32_add), 5, 0, 0),
        "u32_sub": (lambda: capture(E.evaluate_u32_sub), 5, 0, 0),
        "u32_tri_add_carry_as_chunk": (lambda: capture(E.evaluate_u32_tri_add_carry_as_chunk), 17, 0, 0),
        "u8x4_fma": (lambda: capture(E.evaluate_u8x4_fma), 26, 0, 0),
        "fma_in_extension": (lambda: capture(E.evaluate_fma_in_extension), 8, 4, 0),
        "matrix_multiplication_poseidon2_external": (lambda: capture(E.evaluate_matrix_multiplication, E.poseidon2_external_matrix()), 24, 0, 0),
        "matrix_multiplication_poseidon2_inner": (lambda: capture(E.evaluate_matrix_multiplication, E.poseidon2_inner_matrix()), 24, 0, 0),
        "matrix_multiplication_host_matrix": (lambda: capture(E.evaluate_matrix_multiplication, MATRIX), 24, 0, 0),
        "poseidon2_flattened": (lambda: capture(E.evaluate_poseidon2_flattened, 0), 130, 0, 0),
    }


def _convert(cap, dense):
    OPS = {"Add": GP.OP_ADD, "Sub": GP.OP_SUB, "Mul": GP.OP_MUL, "Double": GP.OP_DOUBLE, "Negate": GP.OP_NEGATE,
           "Square": GP.OP_SQUARE, "Inverse": GP.OP_INVERSE}
    KIND = {"var": GP.IDX_VARIABLE, "wit": GP.IDX_WITNESS, "const": GP.IDX_CONSTANT_POLY}
    values, vindex, rename = [], {}, {}

    def ix(idx, define=False):
        k, i = idx
        if k == "tmp":
            if not dense:
                return (GP.IDX_TEMPORARY, i)
            if define:
                rename.setdefault(i, len(rename))
            return (GP.IDX_TEMPORARY, rename[i])
        if k == "value":
            if i not in vindex:
                vindex[i] = len(values)
                values.append(i)
            return (GP.IDX_VALUE, vindex[i])
        return (KIND[k], i)
    relations = []
    for dst, rel in cap.relations:
        a = ix(rel[1])
        b = ix(rel[2]) if len(rel) > 2 else (0, 0)
        relations.append((OPS[rel[0]], ix(dst, True)[1], a, b))
    writes = [ix(w) for w in cap.writes]
    n_tmp = len(rename) if dense else max([r[1] for r in relations], default=-1) + 1
    return GP.GateProgram(relations, values, writes, n_tmp)


def to_program(cap):
    """rust/prove_hip.rs::OwnedProgram::from_capture in Python: temporaries renumbered densely in definition order."""
    return _convert(cap, True)


def to_program_raw(cap):
    """The capture with the reference's own temporary numbers (sparse, starting wherever the process-wide counter stood)."""
    return _convert(cap, False)
