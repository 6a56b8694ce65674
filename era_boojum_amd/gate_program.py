"""Gate op lists for seam S3 (include/boojum_hip.h, bj_gate_program): what the reference's gpu_synthesizer records when it runs
a GateConstraintEvaluator over GpuSynthesizerFieldLike (src/gpu_synthesizer/mod.rs:136-352, 354-444).

`EvaluatorContext` is that recording field: every arithmetic call emits ONE relation into a fresh temporary, with the trait's
default `mul_and_accumulate_into` (a product, then a sum: field_like.rs:71-75), `small_pow` (:78-106) and `pow_u64`.  The
`evaluate_*` functions below are the `evaluate_once` bodies of src/cs/gates/*.rs written against it CALL BY CALL, so a
program built here is the list a Rust host would hand over for that evaluator (up to the numbering of temporaries, which the
library does not depend on: csrc/gate_canon.h).  gate_codegen.py generates the build-time kernels from exactly these lists.

    b = GateProgramBuilder()
    evaluate_fma(b)                     # FmaGateInBaseFieldWithoutConstant
    program = b.build()

A formula can also be written with `+ - *` on the traced values (tests do that for independent restatements)."""
import ctypes as C

OP_ADD, OP_DOUBLE, OP_SUB, OP_NEGATE, OP_MUL, OP_SQUARE, OP_INVERSE = 1, 2, 3, 4, 5, 6, 7
IDX_VARIABLE, IDX_WITNESS, IDX_CONSTANT_POLY, IDX_TEMPORARY, IDX_VALUE = 0, 1, 2, 3, 4
P = (1 << 64) - (1 << 32) + 1


class _GateIndex(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("index", C.c_uint32)]


class _GateRelation(C.Structure):
    _fields_ = [("op", C.c_uint32), ("dst", C.c_uint32), ("a", _GateIndex), ("b", _GateIndex)]


class _GateProgram(C.Structure):
    _fields_ = [("relations", C.POINTER(_GateRelation)), ("num_relations", C.c_uint32), ("values", C.POINTER(C.c_uint64)),
                ("num_values", C.c_uint32), ("writes", C.POINTER(_GateIndex)), ("num_writes", C.c_uint32),
                ("num_temporaries", C.c_uint32)]


class _Val:
    """A traced field element: a column, a constant, or a temporary produced by a recorded relation."""

    def __init__(self, builder, kind, index):
        self.b, self.kind, self.index = builder, kind, index

    def _coerce(self, other):
        return other if isinstance(other, _Val) else self.b.value(other)

    def __add__(self, o): return self.b._emit(OP_ADD, self, self._coerce(o))
    def __radd__(self, o): return self._coerce(o) + self
    def __sub__(self, o): return self.b._emit(OP_SUB, self, self._coerce(o))
    def __rsub__(self, o): return self._coerce(o) - self
    def __mul__(self, o): return self.b._emit(OP_MUL, self, self._coerce(o))
    def __rmul__(self, o): return self._coerce(o) * self
    def __neg__(self): return self.b._emit(OP_NEGATE, self)
    def double(self): return self.b._emit(OP_DOUBLE, self)
    def square(self): return self.b._emit(OP_SQUARE, self)
    def inverse(self): return self.b._emit(OP_INVERSE, self)


class EvaluatorContext:
    """PrimeFieldLike for GpuSynthesizerFieldLike (gpu_synthesizer/mod.rs:232-352) + TraceSource / EvaluationDestination of
    the capture (:55-101).  `x.op_assign(&y, ctx)` of the reference reads `x = c.op(x, y)` here.  Subclasses provide
    `_emit(op, a, b=None) -> value`, `var / wit / const_poly / value` and `push`."""

    def zero(self): return self.value(0)
    def one(self): return self.value(1)
    def minus_one(self): return self.value(P - 1)
    def add(self, x, y): return self._emit(OP_ADD, x, y)
    def sub(self, x, y): return self._emit(OP_SUB, x, y)
    def mul(self, x, y): return self._emit(OP_MUL, x, y)
    def square(self, x): return self._emit(OP_SQUARE, x)
    def negate(self, x): return self._emit(OP_NEGATE, x)
    def double(self, x): return self._emit(OP_DOUBLE, x)
    def inverse(self, x): return self._emit(OP_INVERSE, x)

    def mul_and_accumulate_into(self, acc, a, b):          # field_like.rs:71-75: tmp = a; tmp *= b; acc += tmp
        return self.add(acc, self.mul(a, b))

    def small_pow(self, x, n):                              # field_like.rs:78-106
        if n == 3:
            return self.mul(self.square(x), x)
        if n == 5:
            return self.mul(self.square(self.square(x)), x)
        if n == 7:
            pow2 = self.square(x)
            return self.mul(self.mul(self.square(pow2), pow2), x)
        raise NotImplementedError("small_pow(%d): the reference has 3, 5 and 7" % n)


class GateProgramBuilder(EvaluatorContext):
    def __init__(self):
        self.relations, self.values, self.writes, self.n_tmp = [], [], [], 0

    def var(self, i): return _Val(self, IDX_VARIABLE, i)
    def wit(self, i): return _Val(self, IDX_WITNESS, i)          # a non-copiable witness column (TraceSource::get_witness_value)
    def const_poly(self, i): return _Val(self, IDX_CONSTANT_POLY, i)

    def value(self, x):
        x = int(x) % P
        if x not in self.values:
            self.values.append(x)
        return _Val(self, IDX_VALUE, self.values.index(x))

    def _emit(self, op, a, b=None):
        dst = self.n_tmp
        self.n_tmp += 1
        self.relations.append((op, dst, (a.kind, a.index), (b.kind, b.index) if b is not None else (0, 0)))
        return _Val(self, IDX_TEMPORARY, dst)

    def push(self, v):
        """push_evaluation_result: `v` is a quotient term of one repetition."""
        self.writes.append((v.kind, v.index))

    def build(self):
        """One temporary per recorded relation, numbered densely in definition order — what rust/prove_hip.rs
        `OwnedProgram::from_capture` makes of a GPUDataCapture.  Slot allocation by live range, common subexpressions and the
        choice of a kernel are the library's business (csrc/gate_canon.h), whatever the numbering."""
        return GateProgram(self.relations, self.values, self.writes, self.n_tmp)


class GateProgram:
    def __init__(self, relations, values, writes, num_temporaries):
        self.relations, self.values, self.writes, self.num_temporaries = list(relations), list(values), list(writes), num_temporaries
        self._rel = (_GateRelation * max(1, len(relations)))()
        for i, (op, dst, a, b) in enumerate(relations):
            self._rel[i] = _GateRelation(op, dst, _GateIndex(*a), _GateIndex(*b))
        self._val = (C.c_uint64 * max(1, len(values)))(*values)
        self._wr = (_GateIndex * max(1, len(writes)))(*[_GateIndex(k, i) for k, i in writes])
        self.struct = _GateProgram(self._rel, len(relations), self._val, len(values), self._wr, len(writes), num_temporaries)

    @property
    def num_terms(self):
        return len(self.writes)

    @property
    def witness_width(self):
        """Witness columns one repetition reads (0 for almost every evaluator)."""
        refs = [r for _, _, a, b in self.relations for r in (a, b)] + list(self.writes)
        return max([i + 1 for k, i in refs if k == IDX_WITNESS], default=0)

    def evaluate(self, var, con, wit=()):
        """Reference semantics in python integers (for tests): var/con/wit = lists of ints; returns the terms."""
        tmp = {}

        def get(ix):
            k, i = ix
            return {IDX_VARIABLE: lambda: var[i], IDX_WITNESS: lambda: wit[i], IDX_CONSTANT_POLY: lambda: con[i],
                    IDX_TEMPORARY: lambda: tmp[i], IDX_VALUE: lambda: self.values[i]}[k]() % P
        for op, dst, a, b in self.relations:
            x = get(a)
            if op == OP_ADD: r = x + get(b)
            elif op == OP_DOUBLE: r = 2 * x
            elif op == OP_SUB: r = x - get(b)
            elif op == OP_NEGATE: r = -x
            elif op == OP_MUL: r = x * get(b)
            elif op == OP_SQUARE: r = x * x
            else: r = pow(x, P - 2, P)
            tmp[dst] = r % P
        return [get(w) for w in self.writes]


def _evaluate_columns(self, var_cols, con_cols, wit_cols=()):
    """The same semantics over whole columns (numpy uint64, era_boojum_amd.field_np): returns one array per term."""
    import numpy as np
    from . import field_np as F
    tmp = {}
    n = len(var_cols[0]) if len(var_cols) else len(con_cols[0])

    def get(ix):
        k, i = ix
        if k == IDX_VARIABLE: return F.canon(np.asarray(var_cols[i], dtype=np.uint64))
        if k == IDX_WITNESS: return F.canon(np.asarray(wit_cols[i], dtype=np.uint64))
        if k == IDX_CONSTANT_POLY: return F.canon(np.asarray(con_cols[i], dtype=np.uint64))
        if k == IDX_TEMPORARY: return tmp[i]
        return np.full(n, self.values[i] % P, dtype=np.uint64)
    for op, dst, a, b in self.relations:
        x = get(a)
        if op == OP_ADD: r = F.add(x, get(b))
        elif op == OP_DOUBLE: r = F.add(x, x)
        elif op == OP_SUB: r = F.sub(x, get(b))
        elif op == OP_NEGATE: r = F.sub(np.zeros(n, dtype=np.uint64), x)
        elif op == OP_MUL: r = F.mul(x, get(b))
        elif op == OP_SQUARE: r = F.mul(x, x)
        else: r = np.array([pow(int(v), P - 2, P) for v in x], dtype=np.uint64)
        tmp[dst] = r
    return [get(w) for w in self.writes]


GateProgram.evaluate_columns = _evaluate_columns


# ---- evaluate_once of the reference's evaluators, call by call (src/cs/gates/*.rs) ----
def evaluate_fma(c):
    """FmaGateInBaseWithoutConstantConstraintEvaluator, fma_gate_without_constant.rs:96-126."""
    a, b, cc, d = (c.var(i) for i in range(4))
    quadratic_term_coeff, linear_term_coeff = c.const_poly(0), c.const_poly(1)      # load_row_shared_constants :78-93
    contribution = c.mul(cc, linear_term_coeff)
    t = c.mul(a, b)
    contribution = c.mul_and_accumulate_into(contribution, quadratic_term_coeff, t)
    contribution = c.sub(contribution, d)
    c.push(contribution)


def evaluate_reduction(c, n=4):
    """ReductionGateConstraintEvaluator<N>, reduction_gate.rs:103-126; the N coefficients are row-shared constants."""
    contribution = c.zero()
    for i in range(n):
        contribution = c.mul_and_accumulate_into(contribution, c.var(i), c.const_poly(i))
    contribution = c.sub(contribution, c.var(n))
    c.push(contribution)


def evaluate_constants_allocator(c):
    """ConstantAllocatorConstraintEvaluator, constant_allocator.rs:107-126."""
    c.push(c.sub(c.var(0), c.const_poly(0)))


def evaluate_boolean(c):
    """BooleanConstraintEvaluator, boolean_allocator.rs:100-121."""
    one = c.one()
    a = c.var(0)
    tmp = c.sub(one, a)
    c.push(c.mul(a, tmp))


def evaluate_selection(c):
    """SelectionGateConstraintEvaluator, selection_gate.rs:100-128."""
    a, b, selector, result = (c.var(i) for i in range(4))
    contribution = c.mul(a, selector)
    tmp = c.sub(c.one(), selector)
    contribution = c.mul_and_accumulate_into(contribution, tmp, b)
    contribution = c.sub(contribution, result)
    c.push(contribution)


def evaluate_parallel_selection(c, n=4):
    """ParallelSelectionGateConstraintEvaluator<N>, parallel_selection.rs:106-136."""
    selector = c.var(0)
    for i in range(n):
        a, b, result = c.var(3 * i + 1), c.var(3 * i + 2), c.var(3 * i + 3)
        contribution = c.mul(a, selector)
        tmp = c.sub(c.one(), selector)
        contribution = c.mul_and_accumulate_into(contribution, tmp, b)
        contribution = c.sub(contribution, result)
        c.push(contribution)


def evaluate_conditional_swap(c, n=1):
    """ConditionalSwapGateConstraintEvaluator<N>, conditional_swap.rs:108-152."""
    selector = c.var(0)
    for i in range(n):
        a, b, result_a, result_b = (c.var(4 * i + k) for k in (1, 2, 3, 4))
        contribution = c.mul(b, selector)                  # if we swap - take B
        tmp = c.sub(c.one(), selector)
        contribution = c.mul_and_accumulate_into(contribution, tmp, a)
        contribution = c.sub(contribution, result_a)
        c.push(contribution)
        contribution = c.mul(a, selector)                  # if we swap - take A
        tmp = c.sub(c.one(), selector)
        contribution = c.mul_and_accumulate_into(contribution, tmp, b)
        contribution = c.sub(contribution, result_b)
        c.push(contribution)


def evaluate_dot_product(c, n=4):
    """DotProductConstraintEvaluator<N>, dot_product_gate.rs:102-131."""
    contribution = c.zero()
    for idx in range(n):
        contribution = c.mul_and_accumulate_into(contribution, c.var(2 * idx), c.var(2 * idx + 1))
    contribution = c.sub(contribution, c.var(2 * n))
    c.push(contribution)


def evaluate_quadratic_combination(c, n=4):
    """QuadraticCombinationConstraintEvaluator<N>, quadratic_combination.rs:97-129."""
    contribution = c.mul(c.var(0), c.var(1))
    for i in range(1, n):
        tmp = c.mul(c.var(2 * i), c.var(2 * i + 1))
        contribution = c.add(contribution, tmp)
    c.push(contribution)


def evaluate_reduction_by_powers(c, n=4):
    """ReductionByPowersGateConstraintEvaluator<N>, reduction_by_powers_gate.rs:103-138."""
    reduction_constants = c.const_poly(0)
    current_constant = c.one()
    contribution = c.zero()
    for idx in range(n):
        if idx != 0:
            current_constant = c.mul(current_constant, reduction_constants)
        tmp = c.mul(c.var(idx), current_constant)
        contribution = c.add(contribution, tmp)
    contribution = c.sub(contribution, c.var(n))
    c.push(contribution)


def evaluate_simple_non_linearity(c, n=7):
    """SimpleNonlinearityGateConstraintEvaluator<N>, simple_non_linearity_with_constant.rs:100-125: (x + c)^N = y."""
    x, y = c.var(0), c.var(1)
    tmp = c.add(x, c.const_poly(0))
    contribution = c.small_pow(tmp, n)
    contribution = c.sub(contribution, y)
    c.push(contribution)


def evaluate_zero_check(c, use_witness_column_for_inversion=False):
    """ZeroCheckEvaluator, zero_check.rs:143-176; with use_witness_column_for_inversion the inverse lives in a non-copiable
    witness column (variables_offset 2, witnesses_offset 1, :76-91)."""
    one = c.one()
    inp, flag = c.var(0), c.var(1)
    inversion_witness = c.wit(0) if use_witness_column_for_inversion else c.var(2)
    contribution = c.mul_and_accumulate_into(flag, inp, inversion_witness)
    contribution = c.sub(contribution, one)
    c.push(contribution)
    c.push(c.mul(inp, flag))


def _evaluate_add_with_carry(c, shift):
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


def evaluate_uintx_add(c):
    """UIntXAddConstraintEvaluator, uintx_add.rs:101-140: the shift 2^N is a row-shared constant."""
    _evaluate_add_with_carry(c, c.const_poly(0))


def evaluate_u32_add(c):
    """U32AddConstraintEvaluator, u32_add.rs:93-131: the shift is the field constant 2^32."""
    _evaluate_add_with_carry(c, c.value(1 << 32))


def evaluate_u32_sub(c):
    """U32SubConstraintEvaluator, u32_sub.rs:91-129: a - b - borrow_in - c + 2^32 * borrow_out = 0."""
    a, b, borrow_in, cc, borrow_out = (c.var(i) for i in range(5))
    contribution = c.sub(a, b)
    contribution = c.sub(contribution, borrow_in)
    contribution = c.sub(contribution, cc)
    tmp = c.mul(c.value(1 << 32), borrow_out)
    contribution = c.add(contribution, tmp)
    c.push(contribution)
    contribution = c.mul(borrow_out, borrow_out)
    contribution = c.sub(contribution, borrow_out)
    c.push(contribution)


def evaluate_u32_tri_add_carry_as_chunk(c):
    """U32TriAddCarryAsChunkConstraintEvaluator, u32_tri_add_carry_as_chunk.rs:105-178 (global constants :78-88)."""
    shift8, shift16, shift24, shift32 = (c.value(1 << k) for k in (8, 16, 24, 32))
    one = c.one()
    v = [c.var(i) for i in range(17)]
    coeffs = [one, shift8, shift16, shift24]
    contribution = c.zero()
    for base in (0, 4, 8):                                  # a, b, c limbs
        for k in range(4):
            contribution = c.mul_and_accumulate_into(contribution, v[base + k], coeffs[k])
    contribution = c.sub(contribution, v[12])
    for k, sh in ((1, shift8), (2, shift16), (3, shift24)):
        tmp = c.mul(v[12 + k], sh)
        contribution = c.sub(contribution, tmp)
    tmp = c.mul(v[16], shift32)
    contribution = c.sub(contribution, tmp)
    c.push(contribution)


def evaluate_u8x4_fma(c):
    """U8x4ConstraintEvaluator, u32_fma.rs:140-299 (global constants :73-125): a*b + c + carry_in = low + 2^32 * high."""
    shift_8, shift_16, shift_24 = (c.value(1 << k) for k in (8, 16, 24))
    minus_one = c.value(P - 1)
    minus_shift_8, minus_shift_16, minus_shift_24, minus_shift_32, minus_shift_40 = (c.value(P - (1 << k)) for k in (8, 16, 24, 32, 40))
    v = [c.var(i) for i in range(26)]
    (a0, a1, a2, a3), (b0, b1, b2, b3), (c0, c1, c2, c3) = v[0:4], v[4:8], v[8:12]
    (carry0, carry1, carry2, carry3), (low0, low1, low2, low3), (high0, high1, high2, high3) = v[12:16], v[16:20], v[20:24]
    product_carry0, product_carry1 = v[24], v[25]
    macc = c.mul_and_accumulate_into
    contribution = c0                                                   # + c
    contribution = macc(contribution, c1, shift_8)
    contribution = macc(contribution, c2, shift_16)
    contribution = macc(contribution, c3, shift_24)
    contribution = c.add(contribution, carry0)                          # + carry_in
    contribution = macc(contribution, carry1, shift_8)
    contribution = macc(contribution, carry2, shift_16)
    contribution = macc(contribution, carry3, shift_24)
    contribution = macc(contribution, low0, minus_one)                  # - low
    contribution = macc(contribution, low1, minus_shift_8)
    contribution = macc(contribution, low2, minus_shift_16)
    contribution = macc(contribution, low3, minus_shift_24)
    contribution = macc(contribution, a0, b0)                           # 0..
    tmp = c.mul(a1, b0)                                                 # 8..
    tmp = macc(tmp, a0, b1)
    contribution = macc(contribution, tmp, shift_8)
    tmp = c.mul(a2, b0)                                                 # 16..
    tmp = macc(tmp, a1, b1)
    tmp = macc(tmp, a0, b2)
    contribution = macc(contribution, tmp, shift_16)
    tmp = c.mul(a3, b0)                                                 # 24..
    tmp = macc(tmp, a2, b1)
    tmp = macc(tmp, a1, b2)
    tmp = macc(tmp, a0, b3)
    contribution = macc(contribution, tmp, shift_24)
    contribution = macc(contribution, product_carry0, minus_shift_32)
    contribution = macc(contribution, product_carry1, minus_shift_40)
    c.push(contribution)
    contribution = product_carry0                                       # range 32..64
    contribution = macc(contribution, product_carry1, shift_8)
    contribution = macc(contribution, high0, minus_one)                 # - high
    contribution = macc(contribution, high1, minus_shift_8)
    contribution = macc(contribution, high2, minus_shift_16)
    contribution = macc(contribution, high3, minus_shift_24)
    tmp = c.mul(a3, b1)                                                 # 32..
    tmp = macc(tmp, a2, b2)
    tmp = macc(tmp, a1, b3)
    contribution = c.add(contribution, tmp)
    tmp = c.mul(a3, b2)                                                 # 40..
    tmp = macc(tmp, a2, b3)
    contribution = macc(contribution, tmp, shift_8)
    tmp = c.mul(a3, b3)                                                 # 48..
    contribution = macc(contribution, tmp, shift_16)
    c.push(contribution)


def evaluate_fma_in_extension(c):
    """FmaGateInExtensionWithoutConstantConstraintEvaluator, fma_gate_in_extension_without_constant.rs:116-200: q * a * b + l * c
    = d over F_p[u]/(u^2 - 7); a, b, c, d in variables, q, l row-shared constants, all as (c0, c1) pairs."""
    a_c0, a_c1, b_c0, b_c1, c_c0, c_c1, d_c0, d_c1 = (c.var(i) for i in range(8))
    q_c0, q_c1, l_c0, l_c1 = (c.const_poly(i) for i in range(4))
    non_residue = c.value(7)
    macc = c.mul_and_accumulate_into
    linear_c0 = c.mul(c_c0, l_c0)
    t = c.mul(c_c1, l_c1)
    linear_c0 = macc(linear_c0, t, non_residue)
    linear_c1 = c.mul(c_c0, l_c1)
    linear_c1 = macc(linear_c1, c_c1, l_c0)
    inner_c0 = c.mul(a_c0, b_c0)
    t = c.mul(a_c1, b_c1)
    inner_c0 = macc(inner_c0, t, non_residue)
    inner_c1 = c.mul(a_c0, b_c1)
    inner_c1 = macc(inner_c1, a_c1, b_c0)
    final_c0 = c.mul(inner_c0, q_c0)
    t = c.mul(inner_c1, q_c1)
    final_c0 = macc(final_c0, t, non_residue)
    final_c1 = c.mul(inner_c0, q_c1)
    final_c1 = macc(final_c1, inner_c1, q_c0)
    final_c0 = c.add(final_c0, linear_c0)
    final_c1 = c.add(final_c1, linear_c1)
    c.push(c.sub(final_c0, d_c0))
    c.push(c.sub(final_c1, d_c1))


def evaluate_matrix_multiplication(c, matrix):
    """MatrixMultiplicationEvaluator<N>, matrix_multiplication_gate.rs:98-125: result = M * input, M a global constant."""
    n = len(matrix)
    inp = [c.var(i) for i in range(n)]
    result = [c.var(n + i) for i in range(n)]
    for idx, a in enumerate(result):
        term = c.zero()
        for b, coeff in zip(inp, matrix[idx]):
            term = c.mul_and_accumulate_into(term, b, c.value(int(coeff)))
        c.push(c.sub(term, a))


def poseidon2_round_constants():
    """The 360 round constants (30 rounds x 12) from the generated data table the kernels are built with
    (csrc/poseidon_rc.inc <- src/implementations/poseidon_goldilocks_params.rs:14-105)."""
    import os
    import re
    txt = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "poseidon_rc.inc")).read()
    vals = [int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]+)ULL", txt)]
    assert len(vals) == 360
    return [vals[12 * r: 12 * r + 12] for r in range(30)]


POSEIDON2_M4 = [[5, 7, 1, 3], [4, 6, 1, 1], [1, 3, 5, 7], [1, 1, 4, 6]]     # poseidon2/params.rs:8-33
POSEIDON2_INNER_SHIFTS = [4, 14, 11, 8, 0, 5, 2, 9, 13, 6, 3, 12]            # :35-36


def poseidon2_external_matrix():
    """EXTERNAL_MDS_MATRIX (poseidon2/params.rs:63-94): block circulant circ(2 M4, M4, M4)."""
    return [[POSEIDON2_M4[r % 4][col % 4] * (2 if r // 4 == col // 4 else 1) for col in range(12)] for r in range(12)]


def poseidon2_inner_matrix():
    """INNER_ROUNDS_MATRIX (poseidon2/params.rs:96-105): all ones, diagonal 2^shift + 1."""
    return [[(1 << POSEIDON2_INNER_SHIFTS[r]) + 1 if r == col else 1 for col in range(12)] for r in range(12)]


def evaluate_poseidon2_flattened(c, num_witness_columns_used=0):
    """Poseidon2RoundFunctionFlattenedEvaluator<F, 8, 12, 4, Poseidon2Goldilocks>, poseidon2.rs:166-391: the permutation over 12
    inputs, 12 outputs and a fresh cell ("degree reset") for every S-box input from the second full round on — witness columns
    first while they last, then copiable ones.  Both linear layers are recorded as the reference records them: dense 12 x 12
    products with the matrices as global constants (:114-147), 288 relations each, ~9.5 k relations in all."""
    rc = poseidon2_round_constants()
    full_rc = rc[:4] + rc[26:]                              # full_round_constants()[round]
    partial_rc = [rc[4 + r][0] for r in range(22)]          # inner_round_constants()[round]
    external_matrix = [[c.value(x) for x in row] for row in poseidon2_external_matrix()]
    inner_matrix = [[c.value(x) for x in row] for row in poseidon2_inner_matrix()]
    off = {"var": 24, "wit": 0}

    def next_cell():
        if off["wit"] < num_witness_columns_used:
            off["wit"] += 1
            return c.wit(off["wit"] - 1)
        off["var"] += 1
        return c.var(off["var"] - 1)

    def matmul(matrix, old_state):
        out = []
        for i in range(12):
            tmp = c.zero()
            for src, coeff in zip(old_state, matrix[i]):
                tmp = c.mul_and_accumulate_into(tmp, src, coeff)
            out.append(tmp)
        return out

    def reset_degree(state):
        for i in range(12):
            cell = next_cell()
            c.push(c.sub(state[i], cell))
            state[i] = cell

    state = [c.var(i) for i in range(12)]
    output = [c.var(12 + i) for i in range(12)]
    for rnd in range(4):
        if rnd != 0:
            reset_degree(state)
        else:
            state = matmul(external_matrix, state)
        for idx in range(12):
            state[idx] = c.small_pow(c.add(state[idx], c.value(full_rc[rnd][idx])), 7)
        state = matmul(external_matrix, state)
    for rnd in range(22):
        state[0] = c.add(state[0], c.value(partial_rc[rnd]))
        cell = next_cell()
        c.push(c.sub(state[0], cell))
        state[0] = c.small_pow(cell, 7)
        state = matmul(inner_matrix, state)
    for k in range(4):
        reset_degree(state)
        for idx in range(12):
            state[idx] = c.small_pow(c.add(state[idx], c.value(full_rc[4 + k][idx])), 7)
        state = matmul(external_matrix, state)
    for src, dst in zip(state, output):
        c.push(c.sub(dst, src))


def _program(evaluate, *args, **kw):
    b = GateProgramBuilder()
    evaluate(b, *args, **kw)
    return b.build()


def fma_program(): return _program(evaluate_fma)
def reduction4_program(): return _program(evaluate_reduction, 4)
def constants_allocator_program(): return _program(evaluate_constants_allocator)
def selection_program(): return _program(evaluate_selection)
def dot_product4_program(): return _program(evaluate_dot_product, 4)
def zero_check_program(use_witness_column_for_inversion=False): return _program(evaluate_zero_check, use_witness_column_for_inversion)
def uintx_add_program(): return _program(evaluate_uintx_add)
def boolean_program(): return _program(evaluate_boolean)
def parallel_selection4_program(): return _program(evaluate_parallel_selection, 4)
def u8x4_fma_program(): return _program(evaluate_u8x4_fma)
def conditional_swap_program(n=1): return _program(evaluate_conditional_swap, n)
def quadratic_combination_program(n=4): return _program(evaluate_quadratic_combination, n)
def reduction_by_powers_program(n=4): return _program(evaluate_reduction_by_powers, n)
def simple_non_linearity_program(n=7): return _program(evaluate_simple_non_linearity, n)
def u32_add_program(): return _program(evaluate_u32_add)
def u32_sub_program(): return _program(evaluate_u32_sub)
def u32_tri_add_carry_as_chunk_program(): return _program(evaluate_u32_tri_add_carry_as_chunk)
def fma_in_extension_program(): return _program(evaluate_fma_in_extension)
def matrix_multiplication_program(matrix): return _program(evaluate_matrix_multiplication, matrix)


def poseidon2_flattened_program(num_witness_columns_used=0):
    """The reference's own capture of the gate (~9.5 k relations)."""
    return _program(evaluate_poseidon2_flattened, num_witness_columns_used)


def poseidon2_flattened_compact_program():
    """The same 118 terms from the structured linear layers (out = M4 (x_b + sum x_b); x_i 2^s_i + sum x): ~3.4 k relations.  A
    restatement for checking the capture above and for the CPU oracle's column-wise evaluation, not what a host sends."""
    rc = poseidon2_round_constants()
    m4, shifts = POSEIDON2_M4, POSEIDON2_INNER_SHIFTS
    b = GateProgramBuilder()

    def ext(st):
        blk = []
        for k in range(3):
            x = st[4 * k: 4 * k + 4]
            blk.append([sum((x[j] * m4[i][j] for j in range(1, 4)), x[0] * m4[i][0]) for i in range(4)])
        sums = [blk[0][i] + blk[1][i] + blk[2][i] for i in range(4)]
        return [blk[k][i] + sums[i] for k in range(3) for i in range(4)]

    def inner(st):
        total = st[0]
        for v in st[1:]:
            total = total + v
        return [st[i] * (1 << shifts[i]) + total for i in range(12)]

    def pow7(x):
        x2 = x.square()
        return x2.square() * (x2 * x)

    state = [b.var(i) for i in range(12)]
    output = [b.var(12 + i) for i in range(12)]
    nxt = 24
    for rnd in range(4):
        if rnd != 0:
            for i in range(12):
                v = b.var(nxt)
                nxt += 1
                b.push(state[i] - v)
                state[i] = v
        else:
            state = ext(state)
        state = [pow7(s + rc[rnd][i]) for i, s in enumerate(state)]
        state = ext(state)
    for rnd in range(22):
        state[0] = state[0] + rc[4 + rnd][0]
        v = b.var(nxt)
        nxt += 1
        b.push(state[0] - v)
        state[0] = pow7(v)
        state = inner(state)
    for k in range(4):
        for i in range(12):
            v = b.var(nxt)
            nxt += 1
            b.push(state[i] - v)
            state[i] = v
        state = [pow7(s + rc[26 + k][i]) for i, s in enumerate(state)]
        state = ext(state)
    for s_, o in zip(state, output):
        b.push(o - s_)
    assert nxt == 130 and len(b.writes) == 118
    return b.build()
