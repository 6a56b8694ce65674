"""ORACLE — TEST INFRASTRUCTURE ONLY.

Verification evaluators of the reference's gates, restated for evaluation at ONE point of F_p^2 (what
`Verifier::verify` does with the openings at z, verifier.rs:1530-1720).  Enough gate types to check the quotient
identity of the reference's golden proof (proof.json / vk.json; inner circuit configured at
src/gadgets/recursion/recursive_verifier.rs:2294-2362), which pins — against reference-produced data — the order of the
alpha powers, the selector-path convention, the copy-permutation and lookup terms and every formula below.

Each evaluator: (principal_width, repetitions(geometry), num_row_shared_constants, per-repetition constant stride,
evaluate(var, con) -> [terms]) where `var` are the variables of one repetition and `con` the constants after the selector
path (row-shared first).  Elements are pairs (c0, c1) of python ints mod p.
"""
from oracle.prover import P, eadd, emul, esub

ONE, ZERO = (1, 0), (0, 0)


def _c(x):
    return (x % P, 0)


def _pow7(x):
    x2 = emul(x, x)
    x3 = emul(x2, x)
    x4 = emul(x2, x2)
    return emul(x4, x3)


# --- constant_allocator.rs:107-126: variable - constant (one constant per repetition)
def ev_constants_allocator(var, con):
    return [esub(var[0], con[0])]


# --- boolean_allocator.rs:86-107: a * (1 - a)
def ev_boolean(var, con):
    return [emul(var[0], esub(ONE, var[0]))]


# --- fma_gate_without_constant.rs:96-126: q*a*b + l*c - d, row-shared constants (q, l)
def ev_fma(var, con):
    a, b, c, d = var[:4]
    t = eadd(emul(c, con[1]), emul(con[0], emul(a, b)))
    return [esub(t, d)]


# --- reduction_gate.rs:103-126: sum_k a_k * c_k - out, four row-shared constants
def ev_reduction4(var, con):
    t = ZERO
    for k in range(4):
        t = eadd(t, emul(var[k], con[k]))
    return [esub(t, var[4])]


# --- selection_gate.rs:86-112: sel*a + (1-sel)*b - out
def ev_selection(var, con):
    a, b, sel, out = var[:4]
    t = eadd(emul(a, sel), emul(esub(ONE, sel), b))
    return [esub(t, out)]


# --- parallel_selection.rs:92-120: one selector, N = 4 (a, b, out) triples
def ev_parallel_selection4(var, con):
    sel = var[0]
    out = []
    for i in range(4):
        a, b, r = var[3 * i + 1], var[3 * i + 2], var[3 * i + 3]
        out.append(esub(eadd(emul(a, sel), emul(esub(ONE, sel), b)), r))
    return out


# --- dot_product_gate.rs:85-113: sum_{i<4} a_i*b_i - out
def ev_dot_product4(var, con):
    t = ZERO
    for i in range(4):
        t = eadd(t, emul(var[2 * i], var[2 * i + 1]))
    return [esub(t, var[8])]


# --- zero_check.rs:143-175 (inversion witness in a copiable column): input*inv + flag - 1 ; input*flag
def ev_zero_check(var, con):
    inp, flag, inv = var[:3]
    return [esub(eadd(flag, emul(inp, inv)), ONE), emul(inp, flag)]


# --- the same evaluator with use_witness_column_for_inversion (zero_check.rs:76-91, 157-161): the inverse comes from a non-copiable
#     witness column (TraceSource::get_witness_value(0)); takes the witness values as a third argument
def ev_zero_check_witness(var, con, wit):
    inp, flag = var[:2]
    return [esub(eadd(flag, emul(inp, wit[0])), ONE), emul(inp, flag)]


# --- uintx_add.rs:96-130: a + b + carry_in - c - 2^N*carry_out ; carry_out^2 - carry_out   (row-shared constant 2^N)
def ev_uintx_add(var, con):
    a, b, cin, c, cout = var[:5]
    t = esub(eadd(eadd(a, b), cin), c)
    t = esub(t, emul(con[0], cout))
    return [t, esub(emul(cout, cout), cout)]


# --- u32_fma.rs:96-280 (U8x4FMAGate): a*b + c + carry_in = low + 2^32*high over 8-bit limbs, two relations
def ev_u8x4_fma(var, con):
    a, b, c, carry = var[0:4], var[4:8], var[8:12], var[12:16]
    low, high = var[16:20], var[20:24]
    pc0, pc1 = var[24], var[25]
    s = [_c(1 << (8 * i)) for i in range(6)]
    ms = [_c(-(1 << (8 * i))) for i in range(6)]
    t = c[0]
    for i in (1, 2, 3):
        t = eadd(t, emul(c[i], s[i]))
    t = eadd(t, carry[0])
    for i in (1, 2, 3):
        t = eadd(t, emul(carry[i], s[i]))
    for i in range(4):
        t = eadd(t, emul(low[i], ms[i]))
    t = eadd(t, emul(a[0], b[0]))
    t = eadd(t, emul(eadd(emul(a[1], b[0]), emul(a[0], b[1])), s[1]))
    t = eadd(t, emul(eadd(eadd(emul(a[2], b[0]), emul(a[1], b[1])), emul(a[0], b[2])), s[2]))
    t = eadd(t, emul(eadd(eadd(eadd(emul(a[3], b[0]), emul(a[2], b[1])), emul(a[1], b[2])), emul(a[0], b[3])), s[3]))
    t = eadd(t, emul(pc0, ms[4]))
    t = eadd(t, emul(pc1, ms[5]))
    u = eadd(pc0, emul(pc1, s[1]))
    for i in range(4):
        u = eadd(u, emul(high[i], ms[i]))
    u = eadd(u, eadd(eadd(emul(a[3], b[1]), emul(a[2], b[2])), emul(a[1], b[3])))
    u = eadd(u, emul(eadd(emul(a[3], b[2]), emul(a[2], b[3])), s[1]))
    u = eadd(u, emul(emul(a[3], b[3]), s[2]))
    return [t, u]


# --- poseidon2.rs:165-410 (Poseidon2FlattenedGate<8,12,4>, no witness columns): the permutation with every S-box output
#     of rounds 1.. taken from a fresh variable ("degree reset"), 118 relations over 130 variables
def _poseidon2_tables():
    import oracle as O
    rc = O.poseidon_round_constants()                     # 30 x 12
    m4 = [[5, 7, 1, 3], [4, 6, 1, 1], [1, 3, 5, 7], [1, 1, 4, 6]]
    ext = [[0] * 12 for _ in range(12)]
    for br in range(3):
        for bc in range(3):
            for i in range(4):
                for j in range(4):
                    ext[4 * br + i][4 * bc + j] = m4[i][j] * (2 if br == bc else 1)
    shifts = [4, 14, 11, 8, 0, 5, 2, 9, 13, 6, 3, 12]
    inner = [[1] * 12 for _ in range(12)]
    for i in range(12):
        inner[i][i] = (1 << shifts[i]) + 1
    full = [rc[i] for i in range(4)] + [rc[4 + 22 + i] for i in range(4)]
    partial = [rc[4 + i][0] for i in range(22)]
    return ext, inner, full, partial


def _matmul(m, st):
    out = []
    for row in m:
        t = ZERO
        for coef, v in zip(row, st):
            t = eadd(t, emul(v, _c(coef)))
        out.append(t)
    return out


def ev_poseidon2_flattened(var, con):
    ext, inner, full, partial = _poseidon2_tables()
    state = list(var[0:12])
    output = list(var[12:24])
    nxt = 24
    terms = []
    for rnd in range(4):
        if rnd != 0:
            for i in range(12):
                v = var[nxt]
                nxt += 1
                terms.append(esub(state[i], v))
                state[i] = v
        else:
            state = _matmul(ext, state)
        state = [_pow7(eadd(s, _c(full[rnd][i]))) for i, s in enumerate(state)]
        state = _matmul(ext, state)
    for rnd in range(22):
        state[0] = eadd(state[0], _c(partial[rnd]))
        v = var[nxt]
        nxt += 1
        terms.append(esub(state[0], v))
        state[0] = _pow7(v)
        state = _matmul(inner, state)
    for k in range(4):
        for i in range(12):
            v = var[nxt]
            nxt += 1
            terms.append(esub(state[i], v))
            state[i] = v
        state = [_pow7(eadd(s, _c(full[4 + k][i]))) for i, s in enumerate(state)]
        state = _matmul(ext, state)
    for s, o in zip(state, output):
        terms.append(esub(o, s))
    assert nxt == 130 and len(terms) == 118
    return terms


def ev_nop(var, con):
    return []


# name -> (principal_width, repetitions(num_gp_columns, num_constant_columns), shared constants, constant stride, terms, fn)
EVALUATORS = {
    "ConstantsAllocatorGate": (1, lambda v, k: k, 0, 1, 1, ev_constants_allocator),
    "BooleanConstraintGate": (1, lambda v, k: v, 0, 0, 1, ev_boolean),
    # bounded_boolean_allocator.rs:74-78: the same evaluator, repetitions capped by max_on_row (the descriptor carries the count)
    "BoundedBooleanConstraintGate": (1, lambda v, k: v, 0, 0, 1, ev_boolean),
    "U8x4FMAGate": (26, lambda v, k: v // 26, 0, 0, 2, ev_u8x4_fma),
    "Poseidon2FlattenedGate": (130, lambda v, k: 1, 0, 0, 118, ev_poseidon2_flattened),
    "DotProductGate<4>": (9, lambda v, k: v // 9, 0, 0, 1, ev_dot_product4),
    "ZeroCheckGate": (3, lambda v, k: v // 3, 0, 0, 2, ev_zero_check),
    "ZeroCheckGate[witness]": (2, lambda v, k: v // 2, 0, 0, 2, ev_zero_check_witness),
    "FmaGateInBaseFieldWithoutConstant": (4, lambda v, k: v // 4, 2, 0, 1, ev_fma),
    "UIntXAddGate": (5, lambda v, k: v // 5, 1, 0, 2, ev_uintx_add),
    "SelectionGate": (4, lambda v, k: v // 4, 0, 0, 1, ev_selection),
    "ParallelSelectionGate<4>": (13, lambda v, k: v // 13, 0, 0, 4, ev_parallel_selection4),
    "NopGate": (0, lambda v, k: 1, 0, 0, 0, ev_nop),
    "ReductionGate<4>": (5, lambda v, k: v // 5, 4, 0, 1, ev_reduction4),
}
