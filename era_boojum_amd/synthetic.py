"""SHA-shaped satisfiable synthetic circuits (input synthesis for benches and tests; not on the proving path).

The reference's SHA-256 bench circuit (src/gadgets/sha256/mod.rs:296-375) cannot be synthesised without its Rust CS
machinery, so this module builds a circuit with the SAME geometry and gate mix and a random
satisfying assignment; prover cost is data-independent, so timings are representative ("SHA-shaped synthetic"):

  * 60 general-purpose variable columns, 4 constant columns in the geometry, max constraint degree 4
  * specialized lookups `UseSpecializedColumnsWithTableIdAsConstant { width 4, 8 repetitions, share_table_id }`
    -> 32 more variable columns, one table-id constant column, one multiplicity column, 5 table setup columns
  * gates over general-purpose columns, in the bench's configuration order: ConstantsAllocatorGate,
    FmaGateInBaseFieldWithoutConstant, ReductionGate<4>, NopGate (src/cs/gates/*.rs)
  * selector tree computed with the reference's own placement algorithm (setup.rs:504-728, 1346-1572), which for this
    gate set yields paths FMA=[1], Reduction=[0,1], ConstantsAllocator=[0,0,1], Nop=[0,0,0], 7 constant columns for
    gates (+1 table id = 8) and quotient degree 4
  * sigma polynomials per SURVEY appendix A.1 (setup.rs:24-75, 419-503): sigma = id except on linked cells
  * tables TriXor / Ch / Maj / Split<1> / Split<2> over `table_bits`-bit limbs (4 in the real bench, smaller in tests)
"""
from dataclasses import dataclass, field

import numpy as np

from . import field_np as F

P = F.P

GATE_CONSTANT_ALLOCATOR, GATE_FMA, GATE_REDUCTION4, GATE_NOP = 1, 2, 3, 4
GATE_POSEIDON2_FLATTENED = 6   # the hand-written evaluator of Poseidon2FlattenedGate (csrc/gate_poseidon2.hip)
GATE_PROGRAM = 5     # evaluated from an op list (seam S3): GateDesc.program is an era_boojum_amd.gate_program.GateProgram


@dataclass
class GateDesc:
    kind: int
    name: str
    degree: int              # max_constraint_degree
    num_constants: int       # num_required_constants_in_geometry
    principal_width: int     # variables per repetition
    reps: int                # num_repetitions_in_geometry
    var_stride: int          # per_chunk_offset.variables_offset
    const_stride: int        # per_chunk_offset.constants_offset
    num_terms: int           # quotient terms per repetition
    needs_selector: bool
    path: list = field(default_factory=list)   # selector path, True = constant, False = 1 - constant
    program: object = None   # op list for kind GATE_PROGRAM (and, optionally, for the hand-written kinds)
    wit_stride: int = 0      # per_chunk_offset.witnesses_offset: non-copiable witness columns per repetition
    params: object = None    # the evaluator's own parameters (MatrixMultiplicationGate: the matrix)


def sha_bench_gates(num_gp_vars=60, num_constant_cols=4):
    """Evaluator order of the SHA bench (sha256/mod.rs:340-375)."""
    return [
        GateDesc(GATE_CONSTANT_ALLOCATOR, "ConstantsAllocatorGate", 1, num_constant_cols, 1,
                 min(num_constant_cols, num_gp_vars), 1, 1, 1, True),
        GateDesc(GATE_FMA, "FmaGateInBaseFieldWithoutConstant", 3, 2, 4, num_gp_vars // 4, 4, 0, 1, True),
        GateDesc(GATE_REDUCTION4, "ReductionGate<4>", 2, 4, 5, num_gp_vars // 5, 5, 0, 1, True),
        GateDesc(GATE_NOP, "NopGate", 0, 0, 0, 1, 0, 0, 0, True),
    ]


def extended_gates(num_gp_vars=60, num_constant_cols=4):
    """The bench's gates plus three more evaluator types given only as op lists (seam S3): SelectionGate, ZeroCheckGate
    (two terms per repetition), UIntXAddGate (two terms, one row-shared constant) — src/cs/gates/{selection_gate,
    zero_check,uintx_add}.rs."""
    from . import gate_program as GP
    g = sha_bench_gates(num_gp_vars, num_constant_cols)
    extra = [
        GateDesc(GATE_PROGRAM, "SelectionGate", 2, 0, 4, num_gp_vars // 4, 4, 0, 1, True, program=GP.selection_program()),
        GateDesc(GATE_PROGRAM, "ZeroCheckGate", 2, 0, 3, num_gp_vars // 3, 3, 0, 2, True, program=GP.zero_check_program()),
        GateDesc(GATE_PROGRAM, "UIntXAddGate", 2, 1, 5, num_gp_vars // 5, 5, 0, 2, True, program=GP.uintx_add_program()),
    ]
    return g[:3] + extra + g[3:]


def witness_gates(num_gp_vars=60, num_constant_cols=4, num_witness_cols=5):
    """The bench's gates plus ZeroCheckGate with its inversion witness in a non-copiable witness column
    (use_witness_column_for_inversion = true, zero_check.rs:76-112: 2 variables + 1 witness per repetition)."""
    from . import gate_program as GP
    g = sha_bench_gates(num_gp_vars, num_constant_cols)
    reps = min(num_gp_vars // 2, num_witness_cols)
    zc = GateDesc(GATE_PROGRAM, "ZeroCheckGate[witness]", 2, 0, 2, reps, 2, 0, 2, True, program=GP.zero_check_program(True), wit_stride=1)
    return g[:3] + [zc] + g[3:]


HOST_MATRIX = [[(7 * r + 3 * k + 1) * 65537 % 99991 + 1 for k in range(12)] for r in range(12)]


def host_gates(num_gp_vars=60, num_constant_cols=4, matrix=None):
    """The bench's gates plus an evaluator the library cannot have been built with: MatrixMultiplicationGate<F, 12, PAR> with the
    host's own matrix as its global constant (matrix_multiplication_gate.rs:75-125) — 288 recorded relations that exist only in
    the op list handed over at setup, so its kernel is compiled at run time (csrc/gate_jit.hip)."""
    from . import gate_program as GP
    matrix = matrix or HOST_MATRIX
    g = sha_bench_gates(num_gp_vars, num_constant_cols)
    mm = GateDesc(GATE_PROGRAM, "MatrixMultiplicationGate[host]", 1, 0, 24, num_gp_vars // 24, 24, 0, 12, True,
                  program=GP.matrix_multiplication_program(matrix), params=matrix)
    return g[:3] + [mm] + g[3:]


def recursion_gates(num_gp_vars=130, num_constant_cols=8, poseidon2_as_op_list=False):
    """The evaluators over general-purpose columns of the golden proof's inner circuit (a recursion-layer circuit: 130
    general-purpose columns, 8 x 3 lookup columns, one boolean specialized column = the 155 variable columns of vk.json), in
    its evaluator order; everything but the SHA bench's four hand-written ones is an op list."""
    from . import gate_program as GP
    v = num_gp_vars
    return [
        GateDesc(GATE_CONSTANT_ALLOCATOR, "ConstantsAllocatorGate", 1, num_constant_cols, 1, min(num_constant_cols, v), 1, 1, 1, True),
        GateDesc(GATE_PROGRAM, "U8x4FMAGate", 2, 0, 26, v // 26, 26, 0, 2, True, program=GP.u8x4_fma_program()),
        (GateDesc(GATE_PROGRAM, "Poseidon2FlattenedGate", 7, 0, 130, 1, 130, 0, 118, True, program=GP.poseidon2_flattened_program())
         if poseidon2_as_op_list else GateDesc(GATE_POSEIDON2_FLATTENED, "Poseidon2FlattenedGate", 7, 0, 130, 1, 130, 0, 118, True)),
        GateDesc(GATE_PROGRAM, "DotProductGate<4>", 2, 0, 9, v // 9, 9, 0, 1, True, program=GP.dot_product4_program()),
        GateDesc(GATE_PROGRAM, "ZeroCheckGate", 2, 0, 3, v // 3, 3, 0, 2, True, program=GP.zero_check_program()),
        GateDesc(GATE_FMA, "FmaGateInBaseFieldWithoutConstant", 3, 2, 4, v // 4, 4, 0, 1, True),
        GateDesc(GATE_PROGRAM, "UIntXAddGate", 2, 1, 5, v // 5, 5, 0, 2, True, program=GP.uintx_add_program()),
        GateDesc(GATE_PROGRAM, "SelectionGate", 2, 0, 4, v // 4, 4, 0, 1, True, program=GP.selection_program()),
        GateDesc(GATE_PROGRAM, "ParallelSelectionGate<4>", 2, 0, 13, v // 13, 13, 0, 4, True, program=GP.parallel_selection4_program()),
        GateDesc(GATE_REDUCTION4, "ReductionGate<4>", 2, 4, 5, v // 5, 5, 0, 1, True),
        GateDesc(GATE_NOP, "NopGate", 0, 0, 0, 1, 0, 0, 0, True),
    ]


def recursion_like_circuit(log_n, seed=1, table_bits=2, poseidon2_as_op_list=False):
    """Random satisfiable circuit with the geometry and the gate set of the golden proof's inner circuit: 130 + 24 + 1
    variable columns, width-3 lookups, the Poseidon2 flattened gate (118 terms over 130 variables per row), quotient degree 8."""
    return sha_shaped_circuit(log_n, seed=seed, table_bits=table_bits, num_gp_vars=130, num_constant_cols=8, lookup_width=3,
                              lookup_reps=8, num_public_inputs=2, boolean_columns=1, gates=recursion_gates(130, 8, poseidon2_as_op_list),
                              mix=(0.04, 0.08, 0.12, 0.08, 0.08, 0.12, 0.08, 0.08, 0.08, 0.12), max_allowed_constraint_degree=8)


# ---- selector placement: restatement of TreeNode::try_add_gate / try_find_placement_for_degree (setup.rs:1346-1572) ----
def _stats(node, depth):
    if node[0] == "gate":
        g = node[1]
        return depth + g.degree, g.num_constants + depth
    l, r = _stats(node[1], depth + 1), _stats(node[2], depth + 1)
    return max(l[0], r[0]), max(l[1], r[1])


def _try_add(node, gate, max_deg, max_consts, depth):
    if node is None:
        if depth + gate.degree > max_deg or gate.num_constants > max_consts:
            return None
        return ("gate", gate)
    if node[0] == "gate":
        for cand in (("fork", node, ("gate", gate)), ("fork", ("gate", gate), node)):
            d, c = _stats(cand, depth)
            if d <= max_deg and c <= max_consts:
                return cand
        return None
    new_left = _try_add(node[1], gate, max_deg, max_consts, depth + 1)
    if new_left is not None:
        return ("fork", new_left, node[2])
    new_right = _try_add(node[2], gate, max_deg, max_consts, depth + 1)
    if new_right is not None:
        return ("fork", node[1], new_right)
    return None


def _paths(node, prefix, out):
    if node[0] == "gate":
        out[id(node[1])] = list(prefix)
    else:
        _paths(node[1], prefix + [True], out)
        _paths(node[2], prefix + [False], out)


def place_selectors(gates, num_constant_cols_geometry):
    """compute_selectors_and_constants_placement (setup.rs:504-728).  Returns (max degree incl. selectors, number of
    constant columns needed by gates); fills gate.path."""
    todo = [g for g in gates if g.degree > 0 or g.needs_selector]
    max_degree = max(g.degree for g in todo) - 1
    max_consts = max(g.num_constants for g in todo)
    assert num_constant_cols_geometry >= max_consts
    order = sorted(todo, key=lambda g: (-g.degree, -g.num_constants))   # stable, like sort_by
    target = 1
    while target < max(max_degree, 1):
        target *= 2
    bound_sel = (len(order) - 1).bit_length() if len(order) > 1 else 0   # ceil(log2(#gates))
    for _ in range(4):
        for i in range(bound_sel + 2):
            tree, ok = None, True
            for g in order:
                tree = _try_add(tree, g, target, max_consts + i, 0)
                if tree is None:
                    ok = False
                    break
            if ok:
                paths = {}
                _paths(tree, [], paths)
                for g in order:
                    g.path = paths[id(g)]
                place_selectors.last_tree = tree       # kept for the VerificationKey wire format (wire_format.py)
                return _stats(tree, 0)
        target *= 2
    raise ValueError("no selector placement found")


# ---- non-residues for the copy permutation: make_non_residues (utils.rs:636-688) ----
def non_residues(num_columns, domain_size):
    out = [1]
    current = 1
    seen_pows = []
    while len(out) < num_columns:
        current += 1
        if pow(current, (P - 1) // 2, P) != P - 1:     # not a quadratic non-residue
            continue
        t = pow(current, domain_size, P)
        if t == 1 or t in seen_pows:
            continue
        seen_pows.append(t)
        out.append(current)
    return out


# ---- lookup tables over b-bit limbs (sha256/mod.rs:433-446 uses b = 4) ----
def make_tables(bits):
    m = 1 << bits
    a, b, c = np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij")
    a, b, c = a.reshape(-1), b.reshape(-1), c.reshape(-1)
    mask = m - 1
    tri = np.stack([a, b, c, a ^ b ^ c], axis=1)
    ch = np.stack([a, b, c, ((a & b) ^ (~a & c)) & mask], axis=1)
    maj = np.stack([a, b, c, (a & b) ^ (a & c) ^ (b & c)], axis=1)
    k = np.arange(m)
    s1 = np.stack([k, k & 1, k >> 1, np.zeros_like(k)], axis=1)
    s2 = np.stack([k, k & 3, k >> 2, np.zeros_like(k)], axis=1)
    return [t.astype(np.uint64) for t in (tri, ch, maj, s1, s2)]


TABLE_ID_AS_VARIABLE = 0xFFFFFFFF     # bj_circuit.table_id_col = BJ_TABLE_ID_AS_VARIABLE (include/boojum_hip.h)


@dataclass
class Circuit:
    log_n: int
    num_gp_vars: int
    num_lookup_vars: int
    lookup_width: int
    lookup_reps: int
    gates: list
    num_constant_cols: int          # total, incl. selector columns and the table-id column
    num_constants_for_gates: int
    table_id_col: int
    quotient_degree: int
    variables: np.ndarray           # [V, n] natural order
    multiplicities: np.ndarray      # [1, n]
    sigmas: np.ndarray              # [V, n]
    constants: np.ndarray           # [Kc, n]
    tables: np.ndarray              # [width+1, n]
    non_residues: list
    public_inputs: list             # [(col, row, value)]
    total_tables_len: int
    selector_tree: object = None    # ('gate', GateDesc) | ('fork', left, right); left = constant, right = 1 - constant
    max_allowed_constraint_degree: int = 4
    geometry_constant_cols: int = 4    # CSGeometry::num_constant_columns (the rest of num_constants_for_gates are selector extras)
    specialized_gates: list = field(default_factory=list)   # GateDesc with .program: gates over their own columns after the lookup ones
    witness: np.ndarray = None      # [Wc, n] non-copiable witness columns (WitnessSet::witness, witness.rs:25), or None
    table_id_as_variable: bool = False   # LookupParameters::UseSpecializedColumnsWithTableIdAsVariable (cs/mod.rs:237): every sub-argument
                                         # has lookup_width + 1 variable columns, the last one the table id; no table-id constant column
                                         # (table_id_col == TABLE_ID_AS_VARIABLE, table_ids_column_idxes empty: setup.rs:970-971)

    @property
    def lookup_cols_per_sub(self):
        """specialized_columns_per_subargument (cs/mod.rs:300-312): variable columns of one lookup sub-argument."""
        return self.lookup_width + (1 if self.table_id_as_variable else 0)

    @property
    def num_witness_cols(self):
        return 0 if self.witness is None else int(self.witness.shape[0])

    @property
    def n(self):
        return 1 << self.log_n

    @property
    def num_specialized_vars(self):
        return sum(g.reps * g.var_stride for g in self.specialized_gates)

    @property
    def num_vars(self):
        return self.num_gp_vars + self.num_lookup_vars + self.num_specialized_vars


def sha_shaped_circuit(log_n, seed=42, table_bits=4, mix=(0.05, 0.45, 0.35), num_gp_vars=60, num_constant_cols=4,
                       lookup_width=4, lookup_reps=8, num_public_inputs=2, extended=False, boolean_columns=0, gates=None,
                       max_allowed_constraint_degree=4, num_witness_cols=0, specialized_constant_columns=0,
                       table_id_as_variable=False):
    """Random satisfiable circuit with the SHA bench geometry.  mix = fractions of rows for
    (ConstantsAllocator, FMA, Reduction); the rest are Nop rows.  boolean_columns > 0 adds a BooleanConstraintGate placed
    over that many specialized columns (GatePlacementStrategy::UseSpecializedColumns, boolean_allocator.rs): every row of
    those columns holds a bit, some of them linked to a copy of themselves in another boolean column.
    specialized_constant_columns > 0 adds, after it, a ConstantsAllocatorGate over that many specialized columns with
    share_constants = false: every repetition has its own variable column and its own CONSTANT column (the last constant
    columns, behind the table-id one: evaluator_data.rs:196-238, prover.rs:748-772), cell = constant on every row.
    table_id_as_variable: the lookup argument in its UseSpecializedColumnsWithTableIdAsVariable mode — lookup_width + 1 variable
    columns per sub-argument, the last one holding the id of the table THAT sub-argument looks up on that row (so the eight
    sub-arguments of a row may use eight different tables), and no table-id constant column."""
    n = 1 << log_n
    rng = np.random.default_rng(seed)
    rand_f = lambda shape: rng.integers(0, P, size=shape, dtype=np.uint64)
    custom = gates is not None          # a caller-supplied gate list: `mix` gives one fraction per gate in front of the last one
    if not custom:
        gates = extended_gates(num_gp_vars, num_constant_cols) if extended else sha_bench_gates(num_gp_vars, num_constant_cols)
    if extended and not custom:   # rows: the bench mix squeezed into 60 %, 10 % for each op-list gate, Nop for the rest
        mix = tuple(0.6 * m for m in mix[:3]) + (0.1, 0.1, 0.1)
    max_deg, consts_for_gates = place_selectors(gates, num_constant_cols)
    q = 1
    while q < max_deg - 1:
        q *= 2
    table_id_col = TABLE_ID_AS_VARIABLE if table_id_as_variable else consts_for_gates
    Kc = consts_for_gates + (0 if table_id_as_variable else 1) + specialized_constant_columns
    cps = lookup_width + (1 if table_id_as_variable else 0)
    V = num_gp_vars + cps * lookup_reps + boolean_columns + specialized_constant_columns
    variables = np.zeros((V, n), dtype=np.uint64)
    witness = rand_f((num_witness_cols, n)) if num_witness_cols else None   # unconstrained cells hold anything: they are committed and opened all the same
    constants = np.zeros((Kc, n), dtype=np.uint64)
    swaps = []   # copy cycles of length 2: (col_a, col_b, row slice) — sigma exchanges the two cells' identities

    # --- rows -> gate kinds, in contiguous blocks (prover cost does not depend on the arrangement; slices keep the
    #     generation of a 2^22-row circuit to seconds)
    cuts = [0]
    for m in mix:
        cuts.append(cuts[-1] + int(n * m))
    cuts.append(n)
    for gi, g in enumerate(gates):
        lo, hi = cuts[gi], cuts[gi + 1]
        rows, m = slice(lo, hi), hi - lo
        if m == 0:
            continue
        d = len(g.path)
        for i, bit in enumerate(g.path):
            constants[i, rows] = 1 if bit else 0
        if g.kind == GATE_CONSTANT_ALLOCATOR:
            for r in range(g.reps):
                c = rand_f(m)
                constants[d + r * g.const_stride, rows] = c
                variables[r * g.var_stride, rows] = c
        elif g.kind == GATE_FMA:
            qc, lc = rand_f(m), rand_f(m)
            constants[d, rows], constants[d + 1, rows] = qc, lc
            prev_d = None
            for r in range(g.reps):
                a, b = rand_f(m), rand_f(m)
                c = rand_f(m) if prev_d is None else prev_d      # c_k is a copy of d_{k-1}
                dd = F.fma2(qc, F.mul(a, b), lc, c)
                base = r * g.var_stride
                variables[base, rows], variables[base + 1, rows] = a, b
                variables[base + 2, rows], variables[base + 3, rows] = c, dd
                if prev_d is not None:
                    swaps.append((base + 2, base - 1, rows))
                prev_d = dd
        elif g.name in ("BoundedBooleanConstraintGate", "BooleanConstraintGate"):     # over general-purpose columns: a bit per repetition
            for r in range(g.reps):
                variables[r * g.var_stride, rows] = rng.integers(0, 2, size=m).astype(np.uint64)
        elif g.name.startswith("MatrixMultiplicationGate"):        # result = M * input, M the evaluator's global constant
            for r in range(g.reps):
                base = r * g.var_stride
                x = [rand_f(m) for _ in range(12)]
                for k in range(12):
                    variables[base + k, rows] = x[k]
                for i in range(12):
                    acc = np.zeros(m, dtype=np.uint64)
                    for k in range(12):
                        acc = F.add(acc, F.mul(x[k], np.uint64(int(g.params[i][k]) % P)))
                    variables[base + 12 + i, rows] = acc
        elif g.name == "SelectionGate":
            for r in range(g.reps):
                base = r * g.var_stride
                a, b, sel = rand_f(m), rand_f(m), rng.integers(0, 2, size=m).astype(np.uint64)
                variables[base, rows], variables[base + 1, rows], variables[base + 2, rows] = a, b, sel
                variables[base + 3, rows] = np.where(sel == 1, a, b)
        elif g.name == "ZeroCheckGate[witness]":           # use_witness_column_for_inversion (zero_check.rs:76-91): inverse in a witness column
            for r in range(g.reps):
                base = r * g.var_stride
                x = rand_f(m)
                x[rng.random(m) < 0.3] = 0
                inv = np.array([pow(int(v), P - 2, P) for v in x], dtype=np.uint64)
                variables[base, rows], variables[base + 1, rows] = x, (x == 0).astype(np.uint64)
                witness[r * g.wit_stride, rows] = inv
        elif g.name == "ZeroCheckGate":
            for r in range(g.reps):
                base = r * g.var_stride
                x = rand_f(m)
                x[rng.random(m) < 0.3] = 0                                   # some genuine zeros
                inv = np.array([pow(int(v), P - 2, P) for v in x], dtype=np.uint64)
                variables[base, rows], variables[base + 1, rows], variables[base + 2, rows] = x, (x == 0).astype(np.uint64), inv
        elif g.name == "UIntXAddGate":
            constants[d, rows] = np.uint64(1 << 32)                          # the row-shared shift 2^N, N = 32
            for r in range(g.reps):
                base = r * g.var_stride
                a = rng.integers(0, 1 << 32, size=m, dtype=np.uint64)
                b = rng.integers(0, 1 << 32, size=m, dtype=np.uint64)
                cin = rng.integers(0, 2, size=m).astype(np.uint64)
                tot = a + b + cin
                variables[base, rows], variables[base + 1, rows], variables[base + 2, rows] = a, b, cin
                variables[base + 3, rows], variables[base + 4, rows] = tot & np.uint64(0xFFFFFFFF), tot >> np.uint64(32)
        elif g.name == "DotProductGate<4>":
            for r in range(g.reps):
                base = r * g.var_stride
                v = [rand_f(m) for _ in range(8)]
                for i in range(8):
                    variables[base + i, rows] = v[i]
                variables[base + 8, rows] = F.add(F.fma2(v[0], v[1], v[2], v[3]), F.fma2(v[4], v[5], v[6], v[7]))
        elif g.name == "ParallelSelectionGate<4>":
            for r in range(g.reps):
                base = r * g.var_stride
                sel = rng.integers(0, 2, size=m).astype(np.uint64)
                variables[base, rows] = sel
                for i in range(4):
                    a, b = rand_f(m), rand_f(m)
                    variables[base + 3 * i + 1, rows], variables[base + 3 * i + 2, rows] = a, b
                    variables[base + 3 * i + 3, rows] = np.where(sel == 1, a, b)
        elif g.name == "U8x4FMAGate":                       # a * b + c + carry = low + 2^32 * high on 8-bit limbs (u32_fma.rs)
            for r in range(g.reps):
                base = r * g.var_stride
                a, b, cc, ci = (rng.integers(0, 1 << 32, size=m, dtype=np.uint64) for _ in range(4))
                total = a * b + cc + ci                     # < 2^64
                low, high = total & np.uint64(0xFFFFFFFF), total >> np.uint64(32)
                limbs = lambda x: [(x >> np.uint64(8 * i)) & np.uint64(255) for i in range(4)]
                al, bl = limbs(a), limbs(b)
                cols = al + bl + limbs(cc) + limbs(ci) + limbs(low) + limbs(high)
                # the two product-carry variables: what remains of the low relation after its 32 low bits
                t = (cc + ci + al[0] * bl[0] + ((al[1] * bl[0] + al[0] * bl[1]) << np.uint64(8))
                     + ((al[2] * bl[0] + al[1] * bl[1] + al[0] * bl[2]) << np.uint64(16))
                     + ((al[3] * bl[0] + al[2] * bl[1] + al[1] * bl[2] + al[0] * bl[3]) << np.uint64(24)))
                carry = (t - low) >> np.uint64(32)
                cols += [carry & np.uint64(255), carry >> np.uint64(8)]
                for i, col in enumerate(cols):
                    variables[base + i, rows] = col
        elif g.name == "Poseidon2FlattenedGate":           # one permutation per row: inputs, outputs, the S-box inputs in between
            from .gate_program import poseidon2_round_constants
            rc = poseidon2_round_constants()
            m4 = [[5, 7, 1, 3], [4, 6, 1, 1], [1, 3, 5, 7], [1, 1, 4, 6]]
            shifts = [4, 14, 11, 8, 0, 5, 2, 9, 13, 6, 3, 12]

            # all rows of the gate at once: the state is twelve arrays over the rows
            cst = lambda v: np.uint64(int(v) % P)
            fadd = lambda a, b: F.add(a, b)

            def scaled_sum(cols, coefs):
                acc = None
                for a, k in zip(cols, coefs):
                    t = a if k == 1 else F.mul(a, cst(k))
                    acc = t if acc is None else F.add(acc, t)
                return acc

            def ext(st):
                blk = [[scaled_sum(st[4 * k: 4 * k + 4], m4[i]) for i in range(4)] for k in range(3)]
                sums = [F.add(F.add(blk[0][i], blk[1][i]), blk[2][i]) for i in range(4)]
                return [F.add(blk[k][i], sums[i]) for k in range(3) for i in range(4)]

            def pow7(x):
                x2 = F.mul(x, x)
                return F.mul(F.mul(x2, x2), F.mul(x2, x))
            st = [rand_f(m) for _ in range(12)]
            cells = list(st) + [None] * 12
            st = ext(st)
            for rnd in range(4):
                if rnd:
                    cells += st
                st = ext([pow7(F.add(x, np.full(m, cst(rc[rnd][i]), dtype=np.uint64))) for i, x in enumerate(st)])
            for rnd in range(22):
                st[0] = F.add(st[0], np.full(m, cst(rc[4 + rnd][0]), dtype=np.uint64))
                cells.append(st[0])
                st[0] = pow7(st[0])
                tot = st[0]
                for x in st[1:]:
                    tot = F.add(tot, x)
                st = [F.add(F.mul(st[i], cst(1 << shifts[i])), tot) for i in range(12)]
            for k in range(4):
                cells += st
                st = ext([pow7(F.add(x, np.full(m, cst(rc[26 + k][i]), dtype=np.uint64))) for i, x in enumerate(st)])
            cells[12:24] = st
            assert len(cells) == 130
            for k, col in enumerate(cells):
                variables[k, rows] = col
        elif g.kind == GATE_REDUCTION4:
            cs = [rand_f(m) for _ in range(4)]
            for i in range(4):
                constants[d + i, rows] = cs[i]
            for r in range(g.reps):
                base = r * g.var_stride
                v = [rand_f(m) for _ in range(4)]
                for i in range(4):
                    variables[base + i, rows] = v[i]
                variables[base + 4, rows] = F.add(F.fma2(v[0], cs[0], v[1], cs[1]), F.fma2(v[2], cs[2], v[3], cs[3]))
    # --- lookups: every row looks 8 tuples up in ONE table (shared table id in a constant column)
    tabs = make_tables(table_bits)
    total_len = sum(t.shape[0] for t in tabs)
    if total_len > n:
        raise ValueError("tables (%d rows) do not fit the trace (%d rows): lower table_bits" % (total_len, n))
    tables = np.zeros((lookup_width + 1, n), dtype=np.uint64)
    offs, o = [], 0
    for ti, t in enumerate(tabs):
        tables[:lookup_width, o:o + t.shape[0]] = t.T[:lookup_width]      # narrower lookups use the projection
        tables[lookup_width, o:o + t.shape[0]] = ti + 1         # table ids start at 1 (reference_cs.rs:24)
        offs.append(o)
        o += t.shape[0]
    tid = rng.integers(0, len(tabs), size=n)
    if not table_id_as_variable:
        constants[table_id_col] = (tid + 1).astype(np.uint64)
    sizes = np.array([t.shape[0] for t in tabs])
    offs = np.array(offs)
    mult = np.zeros(n, dtype=np.uint64)
    small = np.ascontiguousarray(tables[:lookup_width, :total_len])
    for rep in range(lookup_reps):
        if table_id_as_variable and rep:               # every sub-argument its own table on every row
            tid = rng.integers(0, len(tabs), size=n)
        pick = offs[tid] + (rng.integers(0, 1 << 62, size=n) % sizes[tid])
        for j in range(lookup_width):
            variables[num_gp_vars + rep * cps + j] = small[j][pick]
        if table_id_as_variable:
            variables[num_gp_vars + rep * cps + lookup_width] = (tid + 1).astype(np.uint64)
        mult += np.bincount(pick, minlength=n).astype(np.uint64)
    specialized = []
    if boolean_columns:
        from .gate_program import boolean_program
        first = num_gp_vars + cps * lookup_reps
        bits = rng.integers(0, 2, size=(boolean_columns, n)).astype(np.uint64)
        if boolean_columns >= 2:                       # column 1 repeats column 0 on the first half: copy constraints
            bits[1, : n // 2] = bits[0, : n // 2]
            swaps.append((first, first + 1, slice(0, n // 2)))
        variables[first:first + boolean_columns] = bits
        specialized.append(GateDesc(GATE_PROGRAM, "BooleanConstraintGate", 2, 0, 1, boolean_columns, 1, 0, 1, False,
                                    program=boolean_program()))
    if specialized_constant_columns:
        from .gate_program import constants_allocator_program
        R = specialized_constant_columns
        vals = rand_f((R, n))
        vals[0, : n // 4] = 5                          # some rows share a value: a copy cycle between two of these cells
        if R >= 2:
            vals[1, : n // 4] = 5
            swaps.append((V - R, V - R + 1, slice(0, n // 4)))
        variables[V - R:] = vals
        constants[Kc - R:] = vals
        specialized.append(GateDesc(GATE_PROGRAM, "ConstantsAllocatorGate", 1, 1, 1, R, 1, 1, 1, False,
                                    program=constants_allocator_program()))
    # --- sigma = id o link,  id(c, r) = k_c * omega^r; linked cells (the FMA chains) exchange identities
    ks = non_residues(V, n)
    om = F.powers(F.omega(log_n), n)
    sigmas = np.empty((V, n), dtype=np.uint64)
    for c in range(V):
        sigmas[c] = F.mul(om, np.uint64(ks[c]))
    for ca, cb, rows in swaps:
        ia, ib = sigmas[ca, rows].copy(), sigmas[cb, rows].copy()
        sigmas[ca, rows], sigmas[cb, rows] = ib, ia
    pubs = []
    for i in range(num_public_inputs):
        col, row = (7 * i + 3) % num_gp_vars, (11 * i + 5) % n
        pubs.append((col, row, int(variables[col, row])))
    return Circuit(log_n, num_gp_vars, cps * lookup_reps, lookup_width, lookup_reps, gates, Kc, consts_for_gates,
                   table_id_col, q, variables, mult.reshape(1, n), sigmas, constants, tables, ks, pubs, total_len,
                   selector_tree=getattr(place_selectors, "last_tree", None), geometry_constant_cols=num_constant_cols,
                   specialized_gates=specialized, max_allowed_constraint_degree=max_allowed_constraint_degree, witness=witness,
                   table_id_as_variable=table_id_as_variable)


def check_satisfied(c: Circuit):
    """Row-level satisfiability (the semantics of check_if_satisfied, satisfiability_test.rs:15): gate terms vanish on
    their rows, linked cells hold equal values, every lookup tuple is in its table, multiplicities are exact."""
    n, V = c.n, c.num_vars
    consts, var = c.constants, c.variables
    sel_rows = {}
    for g in c.gates:
        m = np.ones(n, dtype=bool)
        for i, bit in enumerate(g.path):
            m &= consts[i] == (1 if bit else 0)
        sel_rows[g.name] = m
        d = len(g.path)
        if g.kind == GATE_FMA:
            for r in range(g.reps):
                b = r * g.var_stride
                t = F.sub(F.add(F.mul(consts[d], F.mul(var[b], var[b + 1])), F.mul(consts[d + 1], var[b + 2])), var[b + 3])
                assert not t[m].any(), "FMA unsatisfied"
        elif g.kind == GATE_REDUCTION4:
            for r in range(g.reps):
                b = r * g.var_stride
                acc = np.zeros(n, dtype=np.uint64)
                for i in range(4):
                    acc = F.add(acc, F.mul(var[b + i], consts[d + i]))
                assert not F.sub(acc, var[b + 4])[m].any(), "Reduction unsatisfied"
        elif g.kind == GATE_CONSTANT_ALLOCATOR:
            for r in range(g.reps):
                assert not F.sub(var[r * g.var_stride], consts[d + r * g.const_stride])[m].any(), "ConstAlloc unsatisfied"
        elif g.kind in (GATE_PROGRAM, GATE_POSEIDON2_FLATTENED) and m.any():      # op-list gates: the program itself on the gate's rows
            rows = np.flatnonzero(m)
            prog = g.program
            if prog is None:
                from .gate_program import poseidon2_flattened_compact_program
                prog = poseidon2_flattened_compact_program()
            for r in range(g.reps):
                vcols = [var[r * g.var_stride + k][rows] for k in range(g.principal_width)]
                ccols = [consts[k][rows] for k in range(d + r * g.const_stride, consts.shape[0])]
                wcols = [c.witness[k][rows] for k in range(r * g.wit_stride, c.num_witness_cols)] if c.num_witness_cols else []
                for t in prog.evaluate_columns(vcols, ccols, wcols):
                    assert not t.any(), "%s unsatisfied" % g.name
    assert sum(m.sum() for m in sel_rows.values()) == n, "selector paths must partition the rows"
    col = c.num_gp_vars + c.num_lookup_vars
    ccol = c.num_constant_cols - sum(g.reps * g.const_stride for g in c.specialized_gates)      # their constants: the last columns
    for g in c.specialized_gates:                       # every row, no selector
        for r in range(g.reps):
            ccols = [consts[ccol + r * g.const_stride + k] for k in range(g.const_stride)]
            for t in g.program.evaluate_columns(var[col + r * g.var_stride: col + (r + 1) * g.var_stride], ccols):
                assert not t.any(), "%s over specialized columns unsatisfied" % g.name
        col += g.reps * g.var_stride
        ccol += g.reps * g.const_stride
    # copy constraints: sigma(c, r) = id(c', r')  =>  var[c, r] == var[c', r']
    ks = np.array(c.non_residues, dtype=np.uint64)
    om = F.powers(F.omega(c.log_n), n)
    ids = np.stack([F.mul(ks[col], om) for col in range(V)])
    order = np.argsort(ids.reshape(-1), kind="stable")
    pos = np.searchsorted(ids.reshape(-1)[order], c.sigmas.reshape(-1))
    tgt = order[pos]
    assert np.array_equal(ids.reshape(-1)[tgt], c.sigmas.reshape(-1)), "sigma is not a permutation of the identities"
    assert np.array_equal(np.sort(tgt), np.arange(V * n)), "sigma is not a bijection"
    assert np.array_equal(var.reshape(-1)[tgt], var.reshape(-1)), "copy constraint violated"
    # lookups
    w = c.lookup_width
    enc = lambda cols: sum(cols[j].astype(object) * (1 << (16 * j)) for j in range(len(cols)))
    table_keys = enc(list(c.tables))          # includes the id column
    lut = {}
    for r in range(c.total_tables_len):
        lut[table_keys[r]] = r
    count = np.zeros(n, dtype=np.uint64)
    for rep in range(c.lookup_reps):
        cps = c.lookup_cols_per_sub
        cols = [var[c.num_gp_vars + rep * cps + j] for j in range(cps)] + ([] if c.table_id_as_variable else [consts[c.table_id_col]])
        keys = enc(cols)
        for k in keys:
            count[lut[k]] += 1
    assert np.array_equal(count, c.multiplicities[0]), "multiplicities mismatch"
    return True


_BIG = ("variables", "multiplicities", "sigmas", "constants", "tables", "witness")


def save_circuit(c: Circuit, directory, note=""):
    """The circuit as a directory of plain .npy files (one per big array) + a pickle of everything else: what `bench.py --gpus N`
    uses to synthesise once and let the other ranks MAP the arrays read-only (load_circuit).  Written under a temporary name and
    renamed, so a reader never sees a partial directory."""
    import os
    import pickle
    import shutil
    import copy
    tmp = directory + ".tmp%d" % os.getpid()
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(tmp)
    try:
        small = copy.copy(c)
        for k in _BIG:
            a = getattr(c, k)
            if a is not None:
                np.save(os.path.join(tmp, k + ".npy"), np.ascontiguousarray(a))
            setattr(small, k, None)
        with open(os.path.join(tmp, "circuit.pkl"), "wb") as f:
            pickle.dump({"circuit": small, "note": note, "present": [k for k in _BIG if getattr(c, k) is not None]}, f)
        shutil.rmtree(directory, ignore_errors=True)
        os.rename(tmp, directory)
    except BaseException:
        shutil.rmtree(tmp, ignore_errors=True)      # a full tmpfs must not keep the half-written copy
        raise


def circuit_bytes(c: Circuit):
    return sum(int(getattr(c, k).nbytes) for k in _BIG if getattr(c, k) is not None)


def load_circuit(directory):
    """(Circuit, note) from save_circuit's directory; the big arrays are memory-mapped read-only."""
    import os
    import pickle
    with open(os.path.join(directory, "circuit.pkl"), "rb") as f:
        d = pickle.load(f)
    c = d["circuit"]
    for k in d["present"]:
        setattr(c, k, np.load(os.path.join(directory, k + ".npy"), mmap_mode="r"))
    return c, d["note"]
