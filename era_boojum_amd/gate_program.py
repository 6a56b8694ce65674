"""Gate op lists for seam S3 (include/boojum_hip.h, bj_gate_program): a small tracing builder that records the relations
of a gate formula the way the reference's gpu_synthesizer does (GpuSynthesizerFieldLike + GPUVariablesContext,
src/gpu_synthesizer/mod.rs:136-352): write the evaluator once with `+ - *`, get the op list.

    b = GateProgramBuilder()
    a, bb, c, d = (b.var(i) for i in range(4))
    q, l = b.const_poly(0), b.const_poly(1)
    b.push(q * a * bb + l * c - d)          # FmaGateInBaseFieldWithoutConstant
    program = b.build()
"""
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


class GateProgramBuilder:
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
        """Temporaries are renamed onto as few slots as a linear scan needs (a slot is free again after the last relation
        that reads it; what the writes name stays live): the interpreter has BJ_GATE_PROGRAM_MAX_TEMPORARIES of them, a
        straight trace of a 12 x 12 matrix gate alone would use ~290."""
        last_use = {}
        for i, (op, dst, a, b) in enumerate(self.relations):
            for k, ix in (a, b):
                if k == IDX_TEMPORARY:
                    last_use[ix] = i
        for k, ix in self.writes:
            if k == IDX_TEMPORARY:
                last_use[ix] = len(self.relations)
        slot_of, free, n_slots, relations = {}, [], 0, []

        def rename(ref):
            k, ix = ref
            return (k, slot_of[ix]) if k == IDX_TEMPORARY else ref
        for i, (op, dst, a, b) in enumerate(self.relations):
            ra, rb = rename(a), rename(b)
            for k, ix in {a, b}:                                # operands read for the last time here free their slot:
                if k == IDX_TEMPORARY and last_use[ix] == i:   # the result may take it over (the interpreter reads first)
                    free.append(slot_of[ix])
            if dst not in last_use:                             # a result nobody reads (dead code of a trace)
                last_use[dst] = i
            slot = free.pop() if free else n_slots
            n_slots = max(n_slots, slot + 1)
            slot_of[dst] = slot
            if last_use[dst] == i:
                free.append(slot)
            relations.append((op, slot, ra, rb))
        writes = [rename(w) for w in self.writes]
        return GateProgram(relations, self.values, writes, n_slots)


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


# ---- the evaluators of the SHA bench as op lists (the same formulas quotient.hip hard-codes), and a few more gates ----
def fma_program():
    b = GateProgramBuilder()
    a, bb, c, d = (b.var(i) for i in range(4))
    b.push(b.const_poly(0) * (a * bb) + b.const_poly(1) * c - d)     # fma_gate_without_constant.rs:96-126
    return b.build()


def reduction4_program():
    b = GateProgramBuilder()
    acc = b.var(0) * b.const_poly(0)
    for k in range(1, 4):
        acc = acc + b.var(k) * b.const_poly(k)
    b.push(acc - b.var(4))                                           # reduction_gate.rs:103-126
    return b.build()


def constants_allocator_program():
    b = GateProgramBuilder()
    b.push(b.var(0) - b.const_poly(0))                               # constant_allocator.rs:107-126
    return b.build()


def selection_program():
    b = GateProgramBuilder()
    a, bb, sel, out = (b.var(i) for i in range(4))
    b.push(a * sel + (1 - sel) * bb - out)                           # selection_gate.rs:86-112
    return b.build()


def dot_product4_program():
    b = GateProgramBuilder()
    acc = b.var(0) * b.var(1)
    for i in range(1, 4):
        acc = acc + b.var(2 * i) * b.var(2 * i + 1)
    b.push(acc - b.var(8))                                           # dot_product_gate.rs:85-113
    return b.build()


def zero_check_program(use_witness_column_for_inversion=False):
    """ZeroCheckGate (zero_check.rs:143-175); with use_witness_column_for_inversion the inverse lives in a non-copiable witness
    column (variables_offset 2, witnesses_offset 1, zero_check.rs:76-91)."""
    b = GateProgramBuilder()
    inp, flag = b.var(0), b.var(1)
    inv = b.wit(0) if use_witness_column_for_inversion else b.var(2)
    b.push(flag + inp * inv - 1)                                     # zero_check.rs:143-175
    b.push(inp * flag)
    return b.build()


def uintx_add_program():
    b = GateProgramBuilder()
    a, bb, cin, c, cout = (b.var(i) for i in range(5))
    b.push(a + bb + cin - c - b.const_poly(0) * cout)                # uintx_add.rs:96-130
    b.push(cout.square() - cout)
    return b.build()


def boolean_program():
    b = GateProgramBuilder()
    a = b.var(0)
    b.push(a * (1 - a))                                              # boolean_allocator.rs:86-107
    return b.build()


def parallel_selection4_program():
    b = GateProgramBuilder()
    sel = b.var(0)
    for i in range(4):
        a, bb, r = b.var(3 * i + 1), b.var(3 * i + 2), b.var(3 * i + 3)
        b.push(a * sel + (1 - sel) * bb - r)                         # parallel_selection.rs:92-120
    return b.build()


def u8x4_fma_program():
    """U8x4FMAGate (u32_fma.rs:96-280): a*b + c + carry_in = low + 2^32 * high over 8-bit limbs."""
    b = GateProgramBuilder()
    v = [b.var(i) for i in range(26)]
    a, bb, c, carry, low, high, pc0, pc1 = v[0:4], v[4:8], v[8:12], v[12:16], v[16:20], v[20:24], v[24], v[25]
    sh = lambda i: 1 << (8 * i)
    t = c[0] + c[1] * sh(1) + c[2] * sh(2) + c[3] * sh(3)
    t = t + carry[0] + carry[1] * sh(1) + carry[2] * sh(2) + carry[3] * sh(3)
    for i in range(4):
        t = t - low[i] * sh(i)
    t = t + a[0] * bb[0]
    t = t + (a[1] * bb[0] + a[0] * bb[1]) * sh(1)
    t = t + (a[2] * bb[0] + a[1] * bb[1] + a[0] * bb[2]) * sh(2)
    t = t + (a[3] * bb[0] + a[2] * bb[1] + a[1] * bb[2] + a[0] * bb[3]) * sh(3)
    t = t - pc0 * sh(4) - pc1 * sh(5)
    b.push(t)
    u = pc0 + pc1 * sh(1)
    for i in range(4):
        u = u - high[i] * sh(i)
    u = u + (a[3] * bb[1] + a[2] * bb[2] + a[1] * bb[3])
    u = u + (a[3] * bb[2] + a[2] * bb[3]) * sh(1)
    u = u + (a[3] * bb[3]) * sh(2)
    b.push(u)
    return b.build()


# ---- the remaining evaluators over general-purpose columns (src/cs/gates/*.rs), written once with the tracer ----
def conditional_swap_program(n=1):
    """ConditionalSwapGate<N> (conditional_swap.rs:96-140): selector, then (a, b, result_a, result_b) per pair."""
    b = GateProgramBuilder()
    sel = b.var(0)
    for i in range(n):
        a, bb, ra, rb = (b.var(4 * i + k) for k in (1, 2, 3, 4))
        b.push(bb * sel + (1 - sel) * a - ra)
        b.push(a * sel + (1 - sel) * bb - rb)
    return b.build()


def quadratic_combination_program(n=4):
    """QuadraticCombinationGate<N> (quadratic_combination.rs:85-117): sum a_i * b_i = 0."""
    b = GateProgramBuilder()
    acc = b.var(0) * b.var(1)
    for i in range(1, n):
        acc = acc + b.var(2 * i) * b.var(2 * i + 1)
    b.push(acc)
    return b.build()


def reduction_by_powers_program(n=4):
    """ReductionByPowersGate<N> (reduction_by_powers_gate.rs:96-135): sum var_i * c^i = result, c row-shared."""
    b = GateProgramBuilder()
    c = b.const_poly(0)
    acc, power = b.var(0) * 1, None
    for i in range(1, n):
        power = c if power is None else power * c
        acc = acc + b.var(i) * power
    b.push(acc - b.var(n))
    return b.build()


def simple_non_linearity_program(n=7):
    """SimpleNonlinearityGate<N> (simple_non_linearity_with_constant.rs:96-125): (x + c)^N = y."""
    b = GateProgramBuilder()
    t = b.var(0) + b.const_poly(0)
    acc, base, e = None, t, n
    while e:                                   # small_pow: square and multiply
        if e & 1:
            acc = base if acc is None else acc * base
        e >>= 1
        if e:
            base = base.square()
    b.push(acc - b.var(1))
    return b.build()


def u32_add_program():
    """U32AddGate (u32_add.rs:85-125)."""
    b = GateProgramBuilder()
    a, bb, cin, c, cout = (b.var(i) for i in range(5))
    b.push(a + bb + cin - c - b.value(1 << 32) * cout)
    b.push(cout * cout - cout)
    return b.build()


def u32_sub_program():
    """U32SubGate (u32_sub.rs:85-125): a - b - borrow_in - c + 2^32 * borrow_out = 0."""
    b = GateProgramBuilder()
    a, bb, bin_, c, bout = (b.var(i) for i in range(5))
    b.push(a - bb - bin_ - c + b.value(1 << 32) * bout)
    b.push(bout * bout - bout)
    return b.build()


def u32_tri_add_carry_as_chunk_program():
    """U32TriAddCarryAsChunkGate (u32_tri_add_carry_as_chunk.rs:100-190): three 4x8-bit operands, 8-bit result limbs, carry chunk."""
    b = GateProgramBuilder()
    sh = [1, 1 << 8, 1 << 16, 1 << 24]
    acc = None
    for op in range(3):
        for k in range(4):
            t = b.var(4 * op + k) * sh[k]
            acc = t if acc is None else acc + t
    acc = acc - b.var(12)
    for k in range(1, 4):
        acc = acc - b.var(12 + k) * sh[k]
    b.push(acc - b.var(16) * (1 << 32))
    return b.build()


def fma_in_extension_program():
    """FmaGateInExtensionWithoutConstant (fma_gate_in_extension_without_constant.rs:110-190): q * a * b + l * c = d over
    F_p[u]/(u^2 - 7), all of a, b, c, d, q, l as (c0, c1) pairs; q, l row-shared constants."""
    b = GateProgramBuilder()
    a0, a1, b0, b1, c0, c1, d0, d1 = (b.var(i) for i in range(8))
    q0, q1, l0, l1 = (b.const_poly(i) for i in range(4))
    nr = b.value(7)
    lin0 = c0 * l0 + (c1 * l1) * nr
    lin1 = c0 * l1 + c1 * l0
    in0 = a0 * b0 + (a1 * b1) * nr
    in1 = a0 * b1 + a1 * b0
    f0 = in0 * q0 + (in1 * q1) * nr
    f1 = in0 * q1 + in1 * q0
    b.push(f0 + lin0 - d0)
    b.push(f1 + lin1 - d1)
    return b.build()


def matrix_multiplication_program(matrix):
    """MatrixMultiplicationGate<N> (matrix_multiplication_gate.rs:110-140): result = M * input, M a global constant."""
    n = len(matrix)
    b = GateProgramBuilder()
    for r in range(n):
        acc = None
        for c in range(n):
            t = b.var(c) * int(matrix[r][c])
            acc = t if acc is None else acc + t
        b.push(acc - b.var(n + r))
    return b.build()


def poseidon2_round_constants():
    """The 360 round constants (30 rounds x 12) from the generated data table the kernels are built with
    (csrc/poseidon_rc.inc <- src/implementations/poseidon_goldilocks_params.rs:14-105)."""
    import os
    import re
    txt = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "poseidon_rc.inc")).read()
    vals = [int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]+)ULL", txt)]
    assert len(vals) == 360
    return [vals[12 * r: 12 * r + 12] for r in range(30)]


def poseidon2_flattened_program():
    """Poseidon2FlattenedGate<8, 12, 4> without witness columns (src/cs/gates/poseidon2.rs:165-410): the permutation over
    130 variables — 12 inputs, 12 outputs, and a fresh variable for every S-box input from the second full round on
    ("degree reset") — 118 relations.  ~3.4 k recorded operations on ~150 live slots."""
    rc = poseidon2_round_constants()
    m4 = [[5, 7, 1, 3], [4, 6, 1, 1], [1, 3, 5, 7], [1, 1, 4, 6]]
    shifts = [4, 14, 11, 8, 0, 5, 2, 9, 13, 6, 3, 12]
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
