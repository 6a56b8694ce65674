"""`MemcopySerializable` dumps of the prover's inputs: what a Rust host writes with
`write_into_buffer` can be read here and proved by the HIP prover, without linking anything.

Byte layouts restated from the reference (all integers little-endian):
  Vec<F> / Polynomial            u64 length in field elements, then the raw u64 words
                                 (src/cs/implementations/fast_serialization.rs:139-207, polynomial/mod.rs:95-118)
  Vec<Arc<Polynomial>>           u64 count, then each polynomial             (fast_serialization.rs:17-47)
  SetupBaseStorage               copy_permutation_polys, constant_columns, lookup_tables_columns as above, then bincode of
                                 `table_ids_column_idxes: Vec<usize>` and of `selectors_placement: TreeNode`
                                 (polynomial_storage.rs:77-126); bincode's default: u64 lengths, u32 enum variant tags,
                                 usize as u64, bool as one byte; TreeNode / GateDescription: setup.rs:1378-1396
  WitnessVec                     public_inputs_locations: u64 count + (u64 column, u64 row) pairs; all_values: Vec<F>;
                                 multiplicities: u64 count + u32 each        (witness.rs:29-71)
  DenseVariablesCopyHint         u64 columns, each: u64 length + u64 per cell; bit 63 set = no variable in the cell
                                 (hints/mod.rs:10-61, 104-118; src/cs/mod.rs:44-46, 151-181)
The reference holds no golden bytes for these formats (parity unpinned by vectors): the layouts follow the code cited, the
tests pin them with hand-assembled buffers and round trips.
"""
import struct

import numpy as np

from .synthetic import Circuit, non_residues, _paths, _stats

PLACEHOLDER_BIT = 1 << 63
LOW_U48 = (1 << 48) - 1


class _Reader:
    def __init__(self, buf):
        self.buf, self.pos = memoryview(bytes(buf) if not isinstance(buf, (bytes, bytearray, memoryview)) else buf), 0

    def take(self, n):
        if self.pos + n > len(self.buf):
            raise ValueError("truncated MemcopySerializable buffer")
        out = self.buf[self.pos:self.pos + n]
        self.pos += n
        return out

    def u64(self):
        return struct.unpack("<Q", self.take(8))[0]

    def u32(self):
        return struct.unpack("<I", self.take(4))[0]

    def u8(self):
        return self.take(1)[0]

    def u64_array(self, count):
        return np.frombuffer(self.take(8 * count), dtype="<u8").astype(np.uint64)


# ---- field vectors and polynomials ----
def write_field_vec(values):
    a = np.ascontiguousarray(values, dtype="<u8")
    return struct.pack("<Q", a.size) + a.tobytes()


def read_field_vec(r):
    return r.u64_array(r.u64())


def write_poly_vec(columns):
    return struct.pack("<Q", len(columns)) + b"".join(write_field_vec(c) for c in columns)


def read_poly_vec(r):
    return [read_field_vec(r) for _ in range(r.u64())]


# ---- TreeNode (bincode) ----
def _write_tree(node, gate_index):
    if node is None:
        return struct.pack("<I", 0)
    if node[0] == "gate":
        g = node[1]
        return struct.pack("<IQQQBB", 1, gate_index[id(g)], g.num_constants, g.degree, 1 if g.needs_selector else 0, 0)
    return struct.pack("<I", 2) + _write_tree(node[1], gate_index) + _write_tree(node[2], gate_index)


def _read_tree(r, gates):
    tag = r.u32()
    if tag == 0:
        return None
    if tag == 1:
        idx, num_constants, degree, needs_selector, is_lookup = r.u64(), r.u64(), r.u64(), r.u8(), r.u8()
        if is_lookup:
            raise ValueError("lookup placed through the selector tree (general-purpose-column lookups) is not supported")
        g = gates[idx]
        if (g.num_constants, g.degree) != (num_constants, degree):
            raise ValueError("gate %d (%s): the dump says %d constants / degree %d, the gate description %d / %d"
                             % (idx, g.name, num_constants, degree, g.num_constants, g.degree))
        g.needs_selector = bool(needs_selector)
        return ("gate", g)
    if tag == 2:
        left = _read_tree(r, gates)
        return ("fork", left, _read_tree(r, gates))
    raise ValueError("bad TreeNode variant %d" % tag)


# ---- SetupBaseStorage ----
def write_setup_base(circuit):
    """`SetupBaseStorage::write_into_buffer` for a Circuit."""
    c = circuit
    gate_index = {id(g): i for i, g in enumerate(c.gates)}
    ids = [c.table_id_col] if (c.lookup_reps and not getattr(c, "table_id_as_variable", False)) else []   # setup.rs:970-990
    out = write_poly_vec(list(c.sigmas)) + write_poly_vec(list(c.constants))
    out += write_poly_vec(list(c.tables) if c.lookup_reps else [])
    out += struct.pack("<Q", len(ids)) + b"".join(struct.pack("<Q", i) for i in ids)
    return out + _write_tree(c.selector_tree, gate_index)


def read_setup_base(buf, gates):
    """Returns (sigmas [V, n], constants [Kc, n], tables [w + 1, n] or None, table_ids_column_idxes, selector tree with
    the paths written into `gates` (GateDesc list in evaluator order = gate_idx order)."""
    r = _Reader(buf)
    sig, const, tabs = read_poly_vec(r), read_poly_vec(r), read_poly_vec(r)
    idxes = [r.u64() for _ in range(r.u64())]
    tree = _read_tree(r, gates)
    if r.pos != len(r.buf):
        raise ValueError("%d trailing bytes after SetupBaseStorage" % (len(r.buf) - r.pos))
    n = len(sig[0])
    if any(len(c) != n for c in sig + const + tabs) or n & (n - 1):
        raise ValueError("columns of a SetupBaseStorage must share one power-of-two length")
    paths = {}
    if tree is not None:
        _paths(tree, [], paths)
    for g in gates:
        g.path = paths.get(id(g), [])
    return np.stack(sig), np.stack(const), (np.stack(tabs) if tabs else None), idxes, tree


# ---- WitnessVec and the dense copy hint ----
def write_witness_vec(public_input_locations, all_values, multiplicities):
    out = struct.pack("<Q", len(public_input_locations))
    out += b"".join(struct.pack("<QQ", int(c), int(r)) for c, r in public_input_locations)
    out += write_field_vec(all_values)
    m = np.ascontiguousarray(multiplicities, dtype="<u4")
    return out + struct.pack("<Q", m.size) + m.tobytes()


def read_witness_vec(buf):
    r = _Reader(buf)
    locs = [(r.u64(), r.u64()) for _ in range(r.u64())]
    vals = read_field_vec(r)
    count = r.u64()
    mult = np.frombuffer(r.take(4 * count), dtype="<u4").astype(np.uint32)
    if r.pos != len(r.buf):
        raise ValueError("%d trailing bytes after WitnessVec" % (len(r.buf) - r.pos))
    return locs, vals, mult


def write_variables_hint(var_ids):
    """var_ids [V, n] integers, negative = empty cell."""
    v = np.asarray(var_ids, dtype=np.int64)
    enc = np.where(v >= 0, v.astype(np.uint64), np.uint64(PLACEHOLDER_BIT)).astype("<u8")
    return struct.pack("<Q", v.shape[0]) + b"".join(struct.pack("<Q", v.shape[1]) + enc[c].tobytes() for c in range(v.shape[0]))


def read_variables_hint(buf):
    r = _Reader(buf)
    cols = [r.u64_array(r.u64()) for _ in range(r.u64())]
    if r.pos != len(r.buf):
        raise ValueError("%d trailing bytes after DenseVariablesCopyHint" % (len(r.buf) - r.pos))
    return np.stack(cols)


def variables_from_witness_vec(all_values, hint):
    """witness_set_from_witness_vec (witness.rs:386-443): cell = value of its variable, 0 for a placeholder."""
    empty = (hint & np.uint64(PLACEHOLDER_BIT)) != 0
    idx = (hint & np.uint64(LOW_U48)).astype(np.int64)
    idx[empty] = 0
    out = np.asarray(all_values, dtype=np.uint64)[idx]
    out[empty] = 0
    return out


def multiplicity_column(multiplicities, n):
    """materialize_multiplicities_polynomials (witness.rs:225-272): the per-table counters, concatenated, zero-extended."""
    if len(multiplicities) > n:
        raise ValueError("more multiplicities than rows")
    out = np.zeros((1, n), dtype=np.uint64)
    out[0, :len(multiplicities)] = multiplicities
    return out


def circuit_from_dumps(setup_base, witness_vec, variables_hint, gates, num_gp_vars, lookup_width=0, lookup_reps=0,
                       geometry_constant_cols=4, max_allowed_constraint_degree=4, specialized_gates=()):
    """The prover's inputs from the three dumps a Rust host produces for `prove_cpu_basic` (SetupBaseStorage, WitnessVec,
    DenseVariablesCopyHint) + what is code on the Rust side (the gate list in configuration order, the geometry)."""
    sig, const, tabs, idxes, tree = read_setup_base(setup_base, gates)
    locs, vals, mult = read_witness_vec(witness_vec)
    hint = read_variables_hint(variables_hint)
    V, n = sig.shape
    if hint.shape != (V, n):
        raise ValueError("the copy hint is %s, the setup has %d columns of %d rows" % (hint.shape, V, n))
    # no table-id column among the constants = UseSpecializedColumnsWithTableIdAsVariable: width + 1 variable columns per sub-argument
    tid_var = bool(lookup_reps) and len(idxes) == 0
    cps = lookup_width + (1 if tid_var else 0)
    if V != num_gp_vars + cps * lookup_reps + sum(g.reps * g.var_stride for g in specialized_gates):
        raise ValueError("column count does not match the geometry")
    variables = variables_from_witness_vec(vals, hint)
    max_deg, _ = _stats(tree, 0)
    q = 1
    while q < max_deg - 1:
        q *= 2
    spec_consts = sum(g.reps * g.const_stride for g in specialized_gates)
    if lookup_reps:
        if len(idxes) > 1 or tabs is None or tabs.shape[0] != lookup_width + 1:
            raise ValueError("specialized lookups need at most one table-id column and width + 1 table columns")
        from .synthetic import TABLE_ID_AS_VARIABLE
        table_id_col = TABLE_ID_AS_VARIABLE if tid_var else idxes[0]
        consts_for_gates = const.shape[0] - spec_consts - (0 if tid_var else 1)
        total_len = int(np.count_nonzero(tabs[lookup_width]))
        multiplicities = multiplicity_column(mult, n)
    else:
        table_id_col, consts_for_gates, total_len = const.shape[0], const.shape[0], 0
        tabs = np.zeros((lookup_width + 1, n), dtype=np.uint64)
        multiplicities = np.zeros((1, n), dtype=np.uint64)
    pubs = [(c, r, int(variables[c, r])) for c, r in locs]
    return Circuit(n.bit_length() - 1, num_gp_vars, cps * lookup_reps, lookup_width, lookup_reps, gates, const.shape[0],
                   consts_for_gates, table_id_col, q, variables, multiplicities, sig, const, tabs, non_residues(V, n), pubs,
                   total_len, selector_tree=tree, max_allowed_constraint_degree=max_allowed_constraint_degree,
                   geometry_constant_cols=geometry_constant_cols, specialized_gates=list(specialized_gates),
                   table_id_as_variable=tid_var)
