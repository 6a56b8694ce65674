"""The reference's SHA-256 bench circuit, synthesised for real (input synthesis for benches and tests; not on the
proving path).  SURVEY.md §8(f)-2.

`sha256_circuit(message)` builds what `prove_sha256` builds before it calls the prover (src/gadgets/sha256/mod.rs:
296-487): every input byte allocated and range-checked (`UInt8::allocate_checked`, src/gadgets/u8/mod.rs:68-120, the
4x4x4-table branch), the padded message hashed by the `sha256` gadget (src/gadgets/sha256/mod.rs:35-106) with the round
function of src/gadgets/sha256/round_function.rs:54-297 — 4-bit chunk decompositions, `TriXor4` / `Ch4` / `Maj4` /
`Split4BitChunk<1|2>` lookups, `ReductionGate<4>` and `FmaGateInBaseFieldWithoutConstant` recompositions, constants
through `ConstantsAllocatorGate` — and the rows laid out by the reference's placement rules:

  * a gate over the general-purpose columns joins the half-filled row of its (type, row-shared constants) or opens the
    next free row (`find_next_gate`, src/cs/gates/mod.rs:176-197; the gates' `add_to_cs`); a full row holds 15 FMA /
    12 Reduction / 4 ConstantsAllocator instances; unfilled slots stay empty (zeros satisfy all three gates)
  * a lookup joins the half-filled row of its table or opens the next free row of the 8x4 specialized columns, whose
    constant column holds the table id (`find_next_lookup_gate_specialized`, src/cs/gates/mod.rs:309-341,
    src/cs/implementations/lookup_placement.rs:112-208)
  * finalisation (`pad_and_shrink`, src/cs/implementations/setup.rs:99-371): size = next power of two of
    max(rows + 1, total table length, lookup rows); half-filled lookup rows and all remaining rows are filled with
    row 1 of the table (src/cs/gates/lookup_marker.rs:259-352, src/cs/implementations/lookup_table.rs:401-413), the
    remaining general-purpose rows with NopGate
  * copy-permutation polynomials from the placement (src/cs/implementations/setup.rs:419-503)

Parity: the Rust CS cannot run here, so the layout is checked by what it must satisfy — the digest wired out of the
circuit equals hashlib's, every gate and lookup holds, multiplicities are exact, sigma is a permutation linking equal
values (tests/test_sha256_circuit.py) — and the proof of it is checked by the verifier restatement.

Speed: all full data blocks are structurally identical, so one block is traced symbolically and replayed over all of
them with numpy (a 2^22-row circuit, ~0.5 MB of message, is built in well under a minute).
"""
import numpy as np

from . import field_np as F
from .synthetic import (Circuit, sha_bench_gates, place_selectors, non_residues,
                        GATE_CONSTANT_ALLOCATOR, GATE_FMA, GATE_REDUCTION4, GATE_NOP)

ROUND_CONSTANTS = [
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2,
]
INITIAL_STATE = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]

TRIXOR, CH, MAJ, SPLIT1, SPLIT2 = 1, 2, 3, 4, 5      # table ids in add_lookup_table order (sha256/mod.rs:433-446)
NUM_TABLES = 5
GP_COLUMNS, LOOKUP_WIDTH, LOOKUP_REPS, GEOMETRY_CONSTANTS = 60, 4, 8, 4
CAP = {"fma": GP_COLUMNS // 4, "red": GP_COLUMNS // 5, "const": GEOMETRY_CONSTANTS}
WIDTH = {"fma": 4, "red": 5, "const": 1}

# variable references while tracing: a slot of the current segment, one of its per-repetition external variables, or a
# global variable (a constant)
EXT_TAG, GLB_TAG = 1 << 28, 1 << 29
MAX_GLOBALS = 1 << 12                                 # ids below this are the global variables
u64 = np.uint64
REPLAY_BLOCKS = True     # False: trace every block on its own (tests compare the two)


def sha_tables():
    """Content of the five tables in key-generation order (src/gadgets/tables/{trixor4,ch4,maj4,chunk4bits}.rs)."""
    a, b, c = np.meshgrid(np.arange(16), np.arange(16), np.arange(16), indexing="ij")
    a, b, c = a.reshape(-1), b.reshape(-1), c.reshape(-1)
    k = np.arange(16)
    split = lambda at: np.stack([k, k & ((1 << at) - 1), k >> at, ((k & ((1 << at) - 1)) << (4 - at)) | (k >> at)], axis=1)
    tabs = [np.stack([a, b, c, a ^ b ^ c], axis=1), np.stack([a, b, c, ((a & b) ^ (~a & c)) & 15], axis=1),
            np.stack([a, b, c, (a & b) ^ (a & c) ^ (b & c)], axis=1), split(1), split(2)]
    return [t.astype(np.uint64) for t in tabs]


class _Segment:
    """What one repetition of a stretch of synthesis emits: witness computations, gates, lookups over its own slots."""

    def __init__(self, ext_ids=None):
        self.n_slots = self.n_ext = self.n_inputs = self.n_emit = 0
        self.input_slots = []
        self.wit_ops = []            # (fn, ins, outs)
        self.gates = {}              # key id -> [(emission index, refs)]
        self.lookups = {}            # table id -> [(emission index, refs)]
        self.known_ext_ids = list(ext_ids) if ext_ids is not None else None


class Synthesizer:
    """The slice of the reference's constraint system the SHA-256 gadget uses, recording instead of placing."""

    def __init__(self):
        self.glob_vals = []                    # values of the global variables (constants)
        self.const_vars = {}                   # constant -> global id   (ConstantToVariableMappingTool)
        self.const_rank_values = []            # constants in ConstantsAllocatorGate emission order
        self.keys, self.key_list = {}, []      # (kind, row-shared constants) <-> key id
        self.closed = []                       # closed segments with their instantiation data
        self.next_base = MAX_GLOBALS
        self.order_base = 0
        self.seg = None
        self.vals_chunks = []                  # (base, values[B * n_slots])
        self.recompositions = {}               # 4 global byte variables -> variable id of the word (UInt32RecompositionTooling)

    # ---- segments ----
    def open(self, ext_ids=None):
        assert self.seg is None
        self.seg = _Segment(ext_ids)
        return self.seg

    def key(self, kind, consts):
        k = (kind, tuple(int(c) for c in consts))
        if k not in self.keys:
            self.keys[k] = len(self.key_list)
            self.key_list.append(k)
        return self.keys[k]

    # ---- the ConstraintSystem calls ----
    def alloc(self, count=1):
        s = self.seg
        out = list(range(s.n_slots, s.n_slots + count))
        s.n_slots += count
        return out

    def alloc_input(self):
        """alloc_single_variable_from_witness: the value comes with the repetition's inputs."""
        (v,) = self.alloc()
        self.seg.input_slots.append(v)
        return v

    def ext(self, index):
        self.seg.n_ext = max(self.seg.n_ext, index + 1)
        return EXT_TAG + index

    def witness(self, ins, outs, fn):
        self.seg.wit_ops.append((fn, tuple(ins), tuple(outs)))

    def _emit_gate(self, kid, refs):
        s = self.seg
        s.gates.setdefault(kid, []).append((s.n_emit, tuple(refs)))
        s.n_emit += 1

    def constant(self, value):
        """ConstantsAllocatorGate::allocate_constant (src/cs/gates/constant_allocator.rs:252-293)."""
        value = int(value)
        g = self.const_vars.get(value)
        if g is None:
            assert self.seg.known_ext_ids is not None, "a replayed segment may not meet a new constant"
            g = len(self.glob_vals)
            assert g < MAX_GLOBALS
            self.glob_vals.append(value)
            self.const_vars[value] = g
            self.const_rank_values.append(value)
            self._emit_gate(self.key("const", ()), (GLB_TAG + g,))
        return GLB_TAG + g

    def reduction_gate(self, coeffs, terms, result):
        self._emit_gate(self.key("red", coeffs), list(terms) + [result])

    def reduce_terms(self, coeffs, terms):
        """ReductionGate::reduce_terms (src/cs/gates/reduction_gate.rs:187-225)."""
        (out,) = self.alloc()
        cs_ = [u64(c) for c in coeffs]
        self.witness(terms, [out], lambda a, b, c, d: (a * cs_[0] + b * cs_[1] + c * cs_[2] + d * cs_[3],))
        self.reduction_gate(coeffs, terms, out)
        return out

    def fma_gate(self, c0, c1, a, b, c, d):
        self._emit_gate(self.key("fma", (c0, c1)), (a, b, c, d))

    def compute_fma(self, c0, a, b, c1, c):
        """FmaGateInBaseFieldWithoutConstant::compute_fma (src/cs/gates/fma_gate_without_constant.rs:279-327)."""
        (out,) = self.alloc()
        k0, k1 = u64(c0), u64(c1)
        self.witness([a, b, c], [out], lambda x, y, z: (k0 * x * y + k1 * z,))
        self.fma_gate(c0, c1, a, b, c, out)
        return out

    def enforce_lookup(self, table, refs):
        s = self.seg
        s.lookups.setdefault(table, []).append((s.n_emit, tuple(refs)))
        s.n_emit += 1

    def lookup3(self, table, a, b, c):
        """perform_lookup::<3, 1> (src/cs/implementations/cs.rs:809-858)."""
        (out,) = self.alloc()
        if table == TRIXOR:
            fn = lambda x, y, z: (x ^ y ^ z,)
        elif table == CH:
            fn = lambda x, y, z: ((x & y) ^ (~x & z & u64(15)),)
        else:
            fn = lambda x, y, z: ((x & y) ^ (x & z) ^ (y & z),)
        self.witness([a, b, c], [out], fn)
        self.enforce_lookup(table, (a, b, c, out))
        return out

    # ---- closing a segment: replay it `reps` times ----
    def close(self, reps=1, ext_ids=None, ext_vals=None, inputs=None, chain=None, prev_vals=None):
        """ext_ids / ext_vals [reps, n_ext]: the variables (and their values) the repetitions refer to from outside;
        inputs [reps, n_inputs]: values of alloc_input slots; chain = (ext indices, slots): from the second repetition on
        those externals are the previous repetition's slots.  Returns the segment record (base, n_slots, W)."""
        s, self.seg = self.seg, None
        base = self.next_base
        B, ns, ne = reps, s.n_slots, s.n_ext
        if s.known_ext_ids is not None:
            assert reps == 1
            ext_ids = np.array(s.known_ext_ids[:ne], dtype=np.int64).reshape(1, ne)
        ext_ids = np.zeros((B, 0), dtype=np.int64) if ext_ids is None else np.array(ext_ids, dtype=np.int64).reshape(B, ne)
        if chain is not None and B > 1:
            idx, slots = chain
            rep = np.arange(B - 1, dtype=np.int64)[:, None]
            ext_ids[1:, idx] = base + rep * ns + np.array(slots, dtype=np.int64)[None, :]
        if ext_vals is None:
            ext_vals = self.values_of(ext_ids)
        ext_vals = np.array(ext_vals, dtype=np.uint64).reshape(B, ne)
        # matrices over (rows = slots, externals, globals; columns = repetitions): values W and variable ids ID
        used_globals = sorted({r - GLB_TAG for _, ins, outs in s.wit_ops for r in ins + outs if r >= GLB_TAG} |
                              {r - GLB_TAG for lst in list(s.gates.values()) + list(s.lookups.values())
                               for _, refs in lst for r in refs if r >= GLB_TAG})
        grow = {g: ns + ne + i for i, g in enumerate(used_globals)}

        def row(r):
            return grow[r - GLB_TAG] if r >= GLB_TAG else (ns + (r - EXT_TAG) if r >= EXT_TAG else r)

        W = np.zeros((ns + ne + len(used_globals), B), dtype=np.uint64)
        W[ns:ns + ne] = ext_vals.T
        for g, i in grow.items():
            W[i] = u64(self.glob_vals[g])
        if s.input_slots:
            W[np.array(s.input_slots)] = np.array(inputs, dtype=np.uint64).reshape(B, len(s.input_slots)).T
        for fn, ins, outs in s.wit_ops:
            res = fn(*[W[row(r)] for r in ins])
            for o, v in zip(outs, res):
                W[o] = v
        ID = np.empty((ns + ne + len(used_globals), B), dtype=np.int64)
        ID[:ns] = base + np.arange(B, dtype=np.int64)[None, :] * ns + np.arange(ns, dtype=np.int64)[:, None]
        ID[ns:ns + ne] = ext_ids.T
        for g, i in grow.items():
            ID[i] = g
        rec = dict(base=base, n_slots=ns, reps=B, W=W, ID=ID, order_base=self.order_base, n_emit=s.n_emit,
                   gates={k: (np.array([e for e, _ in lst], dtype=np.int64),
                              np.array([[row(r) for r in refs] for _, refs in lst], dtype=np.int64))
                          for k, lst in s.gates.items()},
                   lookups={t: (np.array([e for e, _ in lst], dtype=np.int64),
                                np.array([[row(r) for r in refs] for _, refs in lst], dtype=np.int64))
                            for t, lst in s.lookups.items()})
        self.closed.append(rec)
        for k, (seg, ref) in list(self.recompositions.items()):
            if seg is s:
                self.recompositions[k] = (None, base + ref)
        self.vals_chunks.append((base, np.ascontiguousarray(W[:ns].T).reshape(-1)))
        self.next_base += B * ns
        self.order_base += B * s.n_emit
        return rec

    def values_of(self, ids):
        ids = np.asarray(ids, dtype=np.int64)
        out = np.zeros(ids.shape, dtype=np.uint64)
        flat, o = ids.reshape(-1), out.reshape(-1)
        for i, v in enumerate(flat):                       # only used for a handful of ids (explicit segments)
            if v < MAX_GLOBALS:
                o[i] = self.glob_vals[v]
            else:
                for base, vals in self.vals_chunks:
                    if base <= v < base + len(vals):
                        o[i] = vals[v - base]
                        break
                else:
                    raise KeyError(v)
        return out


# ---------------------------------------------------------------------------------------------------------------
# the gadget (src/gadgets/sha256/round_function.rs); `cs` is a Synthesizer with an open segment
# ---------------------------------------------------------------------------------------------------------------
_TO_U16 = (1, 1 << 4, 1 << 8, 1 << 12)
_M4 = u64(15)


def _tri_xor_many(cs, a, b, c):
    return [cs.lookup3(TRIXOR, x, y, z) for x, y, z in zip(a, b, c)]


def _uint32_from_4bit_chunks(cs, ch):
    """round_function.rs:316-352"""
    low = cs.reduce_terms(_TO_U16, ch[0:4])
    high = cs.reduce_terms(_TO_U16, ch[4:8])
    one = cs.constant(1)
    return cs.compute_fma(1 << 16, one, high, 1, low)


def _uint32_into_4bit_chunks(cs, x):
    """round_function.rs:354-412"""
    ch = cs.alloc(8)
    cs.witness([x], ch, lambda v: tuple((v >> u64(4 * i)) & _M4 for i in range(8)))
    low = cs.reduce_terms(_TO_U16, ch[0:4])
    high = cs.reduce_terms(_TO_U16, ch[4:8])
    one = cs.constant(1)
    cs.fma_gate(1 << 16, 1, one, high, low, x)
    return ch


def _merge_4bit_chunk(cs, split_at, low, high, swap_output):
    """round_function.rs:551-609"""
    merged = cs.alloc(2)
    s, r = u64(split_at), u64(4 - split_at)
    cs.witness([low, high], merged, lambda lo, hi: (lo | (hi << s), hi | (lo << r)))
    cs.enforce_lookup(SPLIT1 if split_at == 1 else SPLIT2, (merged[0], low, high, merged[1]))
    return merged[1] if swap_output else merged[0]


def _split_and_rotate(cs, x, rotation):
    """round_function.rs:414-549: |4-r|4|4|4|4|4|4|4|r| decomposition, rotation by renumbering, one merge."""
    aligned = cs.alloc(7)
    (dlow,) = cs.alloc()
    (dhigh,) = cs.alloc()
    rm = rotation % 4
    rmu = u64(rm)

    def value(v):
        lowest = v & u64((1 << rm) - 1)
        v = v >> rmu
        mid = tuple((v >> u64(4 * i)) & _M4 for i in range(7))
        return (lowest,) + mid + (v >> u64(28),)

    cs.witness([x], [dlow] + aligned + [dhigh], value)
    shifts = [0, rm] + [rm + 4 * i for i in range(1, 8)]          # bit offsets of dlow, aligned[0..6], dhigh
    t = cs.reduce_terms([1 << shifts[0], 1 << shifts[1], 1 << shifts[2], 1 << shifts[3]],
                        [dlow, aligned[0], aligned[1], aligned[2]])
    t = cs.reduce_terms([1, 1 << shifts[4], 1 << shifts[5], 1 << shifts[6]], [t, aligned[3], aligned[4], aligned[5]])
    zero = cs.constant(0)
    cs.reduction_gate([1, 1 << shifts[7], 1 << shifts[8], 0], [t, aligned[6], dhigh, zero], x)
    if rm == 1:
        merged = _merge_4bit_chunk(cs, 1, dlow, dhigh, True)
    elif rm == 2:
        merged = _merge_4bit_chunk(cs, 2, dhigh, dlow, False)
    else:
        assert rm == 3
        merged = _merge_4bit_chunk(cs, 1, dhigh, dlow, False)
    full = rotation // 4
    result = [None] * 8
    for i, el in enumerate(aligned):
        result[(8 - full + i) % 8] = el
    result[(8 - full - 1) % 8] = merged
    return result, dlow, dhigh


def _range_check_uint32(cs, x):
    """round_function.rs:671-681"""
    ch = _uint32_into_4bit_chunks(cs, x)
    cs.lookup3(TRIXOR, ch[0], ch[1], ch[2])
    cs.lookup3(TRIXOR, ch[3], ch[4], ch[5])
    cs.lookup3(TRIXOR, ch[6], ch[7], ch[0])
    return ch


def _range_check_36_bits(cs, x):
    """round_function.rs:684-762"""
    ch = cs.alloc(9)
    cs.witness([x], ch, lambda v: tuple((v >> u64(4 * i)) & _M4 for i in range(9)))
    low = cs.reduce_terms(_TO_U16, ch[0:4])
    high = cs.reduce_terms(_TO_U16, ch[4:8])
    one = cs.constant(1)
    u32_part = cs.compute_fma(1 << 16, one, high, 1, low)
    cs.fma_gate(1 << 32, 1, one, ch[8], u32_part, x)
    cs.lookup3(TRIXOR, ch[0], ch[1], ch[2])
    cs.lookup3(TRIXOR, ch[3], ch[4], ch[5])
    cs.lookup3(TRIXOR, ch[6], ch[7], ch[8])
    return u32_part


def _split_36_bits_unchecked(cs, x):
    """round_function.rs:765-810"""
    low, high = cs.alloc(2)
    cs.witness([x], [low, high], lambda v: (v & u64(0xFFFFFFFF), v >> u64(32)))
    one = cs.constant(1)
    cs.fma_gate(1 << 32, 1, one, high, low, x)
    return low, high


def _range_check_small(cs, pieces, zero):
    for i in range(0, len(pieces), 3):
        grp = pieces[i:i + 3] + [zero] * (3 - len(pieces[i:i + 3]))
        cs.lookup3(TRIXOR, grp[0], grp[1], grp[2])


def round_function(cs, state, message_block, range_check_final_state):
    """round_function.rs:54-297.  state: 8 refs, replaced in place; returns the 64 4-bit chunks of the new state for
    the last block."""
    expanded = list(message_block) + [None] * 48
    zero, one = cs.constant(0), cs.constant(1)
    unconstrained = []
    for idx in range(16, 64):
        t0 = expanded[idx - 15]
        rot7, _, rot7_high = _split_and_rotate(cs, t0, 7)
        rot18, _, _ = _split_and_rotate(cs, t0, 18)
        shr3 = [rot7[(7 + i) % 8] for i in range(7)] + [rot7_high]
        s0_chunks = _tri_xor_many(cs, rot7, rot18, shr3)
        t1 = expanded[idx - 2]
        rot17, _, _ = _split_and_rotate(cs, t1, 17)
        rot19, _, _ = _split_and_rotate(cs, t1, 19)
        rot10, _, rot10_high = _split_and_rotate(cs, t1, 10)
        shr10 = list(rot10)
        shr10[7], shr10[6], shr10[5] = zero, zero, rot10_high
        s1_chunks = _tri_xor_many(cs, rot17, rot19, shr10)
        s0 = _uint32_from_4bit_chunks(cs, s0_chunks)
        s1 = _uint32_from_4bit_chunks(cs, s1_chunks)
        word = cs.reduce_terms([1, 1, 1, 1], [s0, s1, expanded[idx - 7], expanded[idx - 16]])
        if idx + 2 >= 64:
            expanded[idx] = _range_check_36_bits(cs, word)
        else:
            expanded[idx], high = _split_36_bits_unchecked(cs, word)
            unconstrained.append(high)
    _range_check_small(cs, unconstrained, zero)

    a, b, c, d, e, f, g, h = state
    for rnd in range(64):
        r6, _, _ = _split_and_rotate(cs, e, 6)
        r11, _, _ = _split_and_rotate(cs, e, 11)
        r25, _, _ = _split_and_rotate(cs, e, 25)
        s1 = _uint32_from_4bit_chunks(cs, _tri_xor_many(cs, r6, r11, r25))
        e_ch, f_ch, g_ch = (_uint32_into_4bit_chunks(cs, v) for v in (e, f, g))
        ch = _uint32_from_4bit_chunks(cs, [cs.lookup3(CH, x, y, z) for x, y, z in zip(e_ch, f_ch, g_ch)])
        rc = cs.constant(ROUND_CONSTANTS[rnd])
        tmp1 = cs.reduce_terms([1, 1, 1, 1], [h, s1, ch, rc])
        tmp1 = cs.compute_fma(1, one, tmp1, 1, expanded[rnd])
        t = cs.compute_fma(1, one, tmp1, 1, d)
        new_e = _range_check_36_bits(cs, t)
        r2, _, _ = _split_and_rotate(cs, a, 2)
        r13, _, _ = _split_and_rotate(cs, a, 13)
        r22 = [r2[(i + 5) % 8] for i in range(8)]
        s0 = _uint32_from_4bit_chunks(cs, _tri_xor_many(cs, r2, r13, r22))
        a_ch, b_ch, c_ch = (_uint32_into_4bit_chunks(cs, v) for v in (a, b, c))
        maj = _uint32_from_4bit_chunks(cs, [cs.lookup3(MAJ, x, y, z) for x, y, z in zip(a_ch, b_ch, c_ch)])
        t = cs.reduce_terms([1, 1, 1, 0], [s0, maj, tmp1, zero])
        new_a = _range_check_36_bits(cs, t)
        h, g, f, e, d, c, b, a = g, f, e, new_e, c, b, a, new_a

    final_d = final_h = None
    unchecked = []
    for idx, src in enumerate([a, b, c, d, e, f, g, h]):
        tmp = cs.compute_fma(1, one, state[idx], 1, src)
        tmp, high = _split_36_bits_unchecked(cs, tmp)
        unchecked.append(high)
        if idx == 3:
            final_d = _range_check_uint32(cs, tmp)
        if idx == 7:
            final_h = _range_check_uint32(cs, tmp)
        state[idx] = tmp
    _range_check_small(cs, unchecked, zero)
    if not range_check_final_state:
        return None
    chunks = []
    for idx, el in enumerate(state):
        chunks.append(final_d if idx == 3 else final_h if idx == 7 else _uint32_into_4bit_chunks(cs, el))
    to_check = [v for i in (0, 1, 2, 4, 5, 6) for v in chunks[i]]
    assert len(to_check) == 48
    padded = to_check + [zero] * (38 * 3 - len(to_check))
    for i in range(38):                                         # round_function.rs:281-286 performs 38 lookups
        cs.lookup3(TRIXOR, padded[3 * i], padded[3 * i + 1], padded[3 * i + 2])
    return [v for c8 in chunks for v in c8]


def _allocate_checked_byte(cs):
    """UInt8::allocate_checked with only the 4x4x4 tables present (src/gadgets/u8/mod.rs:68-120, 327-337)."""
    x = cs.alloc_input()
    low, high = cs.alloc(2)
    cs.witness([x], [low, high], lambda v: (v & _M4, v >> u64(4)))
    one = cs.constant(1)
    cs.fma_gate(1 << 4, 1, one, high, low, x)
    cs.lookup3(TRIXOR, low, high, low)
    return x


def _message_words(cs, byte_refs, byte_ids):
    """UInt32::from_be_bytes per 4 bytes (src/gadgets/u32/mod.rs:509-539): words of four already-seen (constant) bytes
    are reused.  byte_ids: global ids when all four bytes are constants, else None."""
    words = []
    for w in range(16):
        le = [byte_refs[4 * w + 3 - j] for j in range(4)]
        ids = None if byte_ids is None else tuple(byte_ids[4 * w + 3 - j] for j in range(4))
        cacheable = ids is not None and all(i is not None for i in ids)
        if cacheable and ids in cs.recompositions:
            seg, ref = cs.recompositions[ids]
            if seg is cs.seg:
                words.append(ref)
            else:                                   # recomposed in an earlier (closed) segment: ref is its variable id
                known = cs.seg.known_ext_ids
                known.append(ref)
                words.append(cs.ext(len(known) - 1))
            continue
        out = cs.reduce_terms([1, 1 << 8, 1 << 16, 1 << 24], le)
        if cacheable:
            cs.recompositions[ids] = (cs.seg, out)
        words.append(out)
    return words


def _sha256_states(padded):
    nb = len(padded) // 64
    states = np.zeros((nb + 1, 8), dtype=np.uint32)
    states[0] = INITIAL_STATE
    if F._NATIVE is not None:
        buf = np.frombuffer(bytes(padded), dtype=np.uint8).copy()
        rc = np.array(ROUND_CONSTANTS, dtype=np.uint32)
        F._NATIVE.synth_sha256_states(buf.ctypes.data, nb, rc.ctypes.data, states.ctypes.data)
        return states
    rotr = lambda x, r: ((x >> r) | (x << (32 - r))) & 0xFFFFFFFF
    for blk in range(nb):
        w = [int.from_bytes(padded[64 * blk + 4 * i: 64 * blk + 4 * i + 4], "big") for i in range(16)]
        for i in range(16, 64):
            s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3)
            s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10)
            w.append((w[i - 16] + s0 + w[i - 7] + s1) & 0xFFFFFFFF)
        a, b, c, d, e, f, g, h = (int(v) for v in states[blk])
        for i in range(64):
            t1 = (h + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + ROUND_CONSTANTS[i] + w[i]) & 0xFFFFFFFF
            t2 = ((rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c))) & 0xFFFFFFFF
            h, g, f, e, d, c, b, a = g, f, e, (d + t1) & 0xFFFFFFFF, c, b, a, (t1 + t2) & 0xFFFFFFFF
        states[blk + 1] = [(int(x) + y) & 0xFFFFFFFF for x, y in zip(states[blk], (a, b, c, d, e, f, g, h))]
    return states


def synthesize(message: bytes):
    """Run the bench's synthesis (sha256/mod.rs:456-470) and return (Synthesizer, digest byte variable ids)."""
    message = bytes(message)
    n_bytes = len(message)
    assert n_bytes >= 1
    cs = Synthesizer()
    data = np.frombuffer(message, dtype=np.uint8).astype(np.uint64)
    # --- input bytes: the first one explicitly (it allocates the constant 1), the others replayed
    cs.open(ext_ids=[])
    _allocate_checked_byte(cs)
    first = cs.close(inputs=data[:1].reshape(1, 1))
    byte_ids = [first["base"] + 0]
    if n_bytes > 1:
        cs.open()
        _allocate_checked_byte(cs)
        rest = cs.close(reps=n_bytes - 1, inputs=data[1:].reshape(-1, 1))
        byte_ids = np.concatenate([byte_ids, rest["base"] + np.arange(n_bytes - 1, dtype=np.int64) * rest["n_slots"]])
    byte_ids = np.asarray(byte_ids, dtype=np.int64)
    # --- padding (sha256/mod.rs:39-63) and the initial state (:67), all constants
    last = n_bytes % 64
    zeros = 64 - 1 - 8 - last if last <= 64 - 1 - 8 else 128 - 1 - 8 - last
    padded = message + b"\x80" + b"\x00" * zeros + (8 * n_bytes).to_bytes(8, "big")
    assert len(padded) % 64 == 0
    num_blocks = len(padded) // 64
    states = _sha256_states(padded)
    cs.open(ext_ids=[])
    pad_refs = [cs.constant(0x80)]
    if zeros > 0:
        z = cs.constant(0)
        pad_refs += [z] * zeros
    pad_refs += [cs.constant(b) for b in (8 * n_bytes).to_bytes(8, "big")]
    iv_refs = [cs.constant(v) for v in INITIAL_STATE]
    cs.close()
    pad_ids = [r - GLB_TAG for r in pad_refs]
    state_ids = [r - GLB_TAG for r in iv_refs]
    full_blocks = n_bytes // 64                      # blocks made of input bytes only
    output = None

    def explicit_block(blk, state_ids):
        """A block traced on its own: the first one (it meets the constants), and those holding padding."""
        known = list(state_ids)
        refs, gids = [], []
        for j in range(64):
            pos = 64 * blk + j
            if pos < n_bytes:
                known.append(int(byte_ids[pos]))
                refs.append(EXT_TAG + len(known) - 1)
                gids.append(None)
            else:
                refs.append(GLB_TAG + pad_ids[pos - n_bytes])
                gids.append(pad_ids[pos - n_bytes])
        cs.open(ext_ids=known)
        cs.seg.n_ext = len(known)
        state = [EXT_TAG + i for i in range(8)]
        words = _message_words(cs, refs, gids)
        is_last = blk == num_blocks - 1
        chunks = round_function(cs, state, words, is_last)
        out_slots = None
        if is_last:                                  # sha256/mod.rs:86-103: the digest bytes from the 4-bit chunks
            one = cs.constant(1)
            out_slots = []
            for w in range(8):
                bytes_le = [cs.compute_fma(1 << 4, one, chunks[8 * w + 2 * j + 1], 1, chunks[8 * w + 2 * j]) for j in range(4)]
                out_slots += bytes_le[::-1]
        rec = cs.close()
        new_state = [rec["base"] + s for s in state]
        got = rec["W"][np.array(state), 0]
        assert np.array_equal(got, states[blk + 1].astype(np.uint64)), "round function witness disagrees with SHA-256"
        return rec, new_state, (None if out_slots is None else [rec["base"] + s for s in out_slots])

    blk = 0
    while blk < num_blocks:
        if blk == 0 or blk >= full_blocks or not REPLAY_BLOCKS:
            rec, state_ids, output = explicit_block(blk, state_ids)
            blk += 1
            continue
        # blocks blk .. full_blocks-1: one trace replayed
        reps = full_blocks - blk
        cs.open()
        state = [cs.ext(i) for i in range(8)]
        refs = [cs.ext(8 + j) for j in range(64)]
        words = _message_words(cs, refs, None)
        round_function(cs, state, words, False)
        ext_ids = np.empty((reps, 72), dtype=np.int64)
        ext_ids[0, :8] = state_ids
        ext_ids[:, 8:] = byte_ids[64 * blk: 64 * full_blocks].reshape(reps, 64)
        ext_vals = np.empty((reps, 72), dtype=np.uint64)
        ext_vals[:, :8] = states[blk: full_blocks].astype(np.uint64)
        ext_vals[:, 8:] = data[64 * blk: 64 * full_blocks].reshape(reps, 64)
        rec = cs.close(reps=reps, ext_ids=ext_ids, ext_vals=ext_vals, chain=(np.arange(8), state))
        got = rec["W"][np.array(state)].T
        assert np.array_equal(got, states[blk + 1: full_blocks + 1].astype(np.uint64)), "replayed blocks disagree with SHA-256"
        state_ids = [rec["base"] + (reps - 1) * rec["n_slots"] + s for s in state]
        blk = full_blocks
    return cs, output



# ---------------------------------------------------------------------------------------------------------------
# placement and finalisation
# ---------------------------------------------------------------------------------------------------------------
def _next_pow2(x):
    return 1 << max(0, int(x - 1).bit_length())


def _assign_rows(cs, which, cap_of):
    """Rows of all instances of `which` ("gates" | "lookups"): an instance joins the half-filled row of its key or opens
    the next free row, in emission order.  Returns (number of rows, per (segment, key): ranks R [reps, count],
    rows_by_chunk per key, totals per key, openers (row -> key))."""
    totals, entries = {}, []
    op_order, op_key, op_chunk = [], [], []
    for rec in cs.closed:
        B = rec["reps"]
        rep = np.arange(B, dtype=np.int64)[:, None]
        for key, (emit, refs) in rec[which].items():
            cap, c = cap_of(key), len(emit)
            R = totals.get(key, 0) + rep * c + np.arange(c, dtype=np.int64)[None, :]
            order = rec["order_base"] + rep * rec["n_emit"] + emit[None, :]
            m = (R % cap) == 0
            op_order.append(order[m])
            op_chunk.append(R[m] // cap)
            op_key.append(np.full(int(m.sum()), key, dtype=np.int64))
            entries.append((rec, key, R))
            totals[key] = totals.get(key, 0) + B * c
    op_order, op_key, op_chunk = np.concatenate(op_order), np.concatenate(op_key), np.concatenate(op_chunk)
    perm = np.argsort(op_order, kind="stable")
    row_key = op_key[perm]
    rows_by_chunk = {}
    rows = np.empty(len(perm), dtype=np.int64)
    rows[perm] = np.arange(len(perm), dtype=np.int64)
    for key, total in totals.items():
        sel = op_key == key
        arr = np.empty(int(sel.sum()), dtype=np.int64)
        arr[op_chunk[sel]] = rows[sel]
        rows_by_chunk[key] = arr
    return len(perm), entries, rows_by_chunk, totals, row_key


def sha256_circuit(message: bytes, return_info=False):
    """The bench's circuit for `message` as a synthetic.Circuit (same container the SHA-shaped generator returns)."""
    cs, output = synthesize(message)
    gates = sha_bench_gates(GP_COLUMNS, GEOMETRY_CONSTANTS)
    max_deg, consts_for_gates = place_selectors(gates, GEOMETRY_CONSTANTS)
    q = 1
    while q < max_deg - 1:
        q *= 2
    by_kind = {g.kind: g for g in gates}
    gate_of = {"const": by_kind[GATE_CONSTANT_ALLOCATOR], "fma": by_kind[GATE_FMA], "red": by_kind[GATE_REDUCTION4]}
    kind_of_key = [k[0] for k in cs.key_list]
    gp_rows, g_entries, g_rows, g_totals, g_row_key = _assign_rows(cs, "gates", lambda k: CAP[kind_of_key[k]])
    lk_rows, l_entries, l_rows, l_totals, l_row_key = _assign_rows(cs, "lookups", lambda t: LOOKUP_REPS)
    tabs = sha_tables()
    total_len = sum(t.shape[0] for t in tabs)
    offs = np.cumsum([0] + [t.shape[0] for t in tabs])[:-1]
    n = _next_pow2(max(gp_rows + 1, total_len))                  # setup.rs:123-141
    if lk_rows > n:                                               # lookup_marker.rs:316-323
        n = _next_pow2(lk_rows)
    log_n = n.bit_length() - 1
    V = GP_COLUMNS + LOOKUP_WIDTH * LOOKUP_REPS
    table_id_col = consts_for_gates
    Kc = consts_for_gates + 1
    var_ids = np.full((V, n), -1, dtype=np.int32)
    constants = np.zeros((Kc, n), dtype=np.uint64)
    # --- general-purpose rows
    for rec, key, R in g_entries:
        kind = kind_of_key[key]
        cap, w = CAP[kind], WIDTH[kind]
        rows, slot = g_rows[key][R // cap], R % cap
        refs = rec["gates"][key][1]
        for j in range(w):
            var_ids[slot * w + j, rows] = rec["ID"][refs[:, j]].T
        if kind == "const":
            d = len(gate_of[kind].path)
            constants[d + slot, rows] = np.array(cs.const_rank_values, dtype=np.uint64)[R]
    for key, (kind, consts) in enumerate(cs.key_list):
        rows = np.flatnonzero(g_row_key == key)
        g = gate_of[kind]
        for i, bit in enumerate(g.path):
            constants[i, rows] = 1 if bit else 0
        for i, cval in enumerate(consts):
            constants[len(g.path) + i, rows] = u64(cval)
    nop = by_kind[GATE_NOP]
    for i, bit in enumerate(nop.path):
        constants[i, gp_rows:] = 1 if bit else 0
    # --- lookup rows, multiplicities
    mult = np.zeros(n, dtype=np.uint64)
    for rec, t, R in l_entries:
        rows, slot = l_rows[t][R // LOOKUP_REPS], R % LOOKUP_REPS
        refs = rec["lookups"][t][1]
        for j in range(LOOKUP_WIDTH):
            var_ids[GP_COLUMNS + slot * LOOKUP_WIDTH + j, rows] = rec["ID"][refs[:, j]].T
        W = rec["W"]
        if t <= MAJ:
            idx = (W[refs[:, 0]] << u64(8)) | (W[refs[:, 1]] << u64(4)) | W[refs[:, 2]]
        else:
            idx = W[refs[:, 0]]
        assert int(idx.max()) < tabs[t - 1].shape[0]
        for j in range(LOOKUP_WIDTH):                       # every tuple is a table row (else the witness is wrong)
            assert np.array_equal(tabs[t - 1][idx.reshape(-1).astype(np.int64), j], W[refs[:, j]].reshape(-1)), \
                "lookup tuple not in table %d" % t
        mult += np.bincount(idx.reshape(-1).astype(np.int64) + int(offs[t - 1]), minlength=n).astype(np.uint64)
    constants[table_id_col, :lk_rows] = l_row_key.astype(np.uint64)
    # padding with row 1 of the table (lookup_table.rs:401-413): half-filled rows with their own table, the rest with table 1
    next_id = cs.next_base
    pad_vals = []
    for t in range(1, NUM_TABLES + 1):
        assert t in l_totals, "every table must be used at least once (lookup_marker.rs:307)"
        placed = (l_totals[t] - 1) % LOOKUP_REPS + 1
        ids = np.arange(next_id, next_id + LOOKUP_WIDTH, dtype=np.int64)
        next_id += LOOKUP_WIDTH
        pad_vals.append(tabs[t - 1][1])
        row = l_rows[t][-1]
        for slot in range(placed, LOOKUP_REPS):
            var_ids[GP_COLUMNS + slot * LOOKUP_WIDTH: GP_COLUMNS + (slot + 1) * LOOKUP_WIDTH, row] = ids
        mult[offs[t - 1] + 1] += u64(LOOKUP_REPS - placed)
    ids = np.arange(next_id, next_id + LOOKUP_WIDTH, dtype=np.int64)
    next_id += LOOKUP_WIDTH
    pad_vals.append(tabs[0][1])
    if lk_rows < n:
        for slot in range(LOOKUP_REPS):
            var_ids[GP_COLUMNS + slot * LOOKUP_WIDTH: GP_COLUMNS + (slot + 1) * LOOKUP_WIDTH, lk_rows:] = ids[:, None]
        constants[table_id_col, lk_rows:] = u64(TRIXOR)
        mult[offs[0] + 1] += u64((n - lk_rows) * LOOKUP_REPS)
    # --- values of all variables
    vals = np.zeros(next_id, dtype=np.uint64)
    vals[:len(cs.glob_vals)] = np.array(cs.glob_vals, dtype=np.uint64)
    for base, chunk in cs.vals_chunks:
        vals[base: base + len(chunk)] = chunk
    vals[cs.next_base:] = np.concatenate(pad_vals)
    if F._NATIVE is not None:
        variables = np.empty((V, n), dtype=np.uint64)
        F._NATIVE.synth_gather_values(var_ids.ctypes.data, vals.ctypes.data, variables.ctypes.data, V * n)
    else:
        variables = np.where(var_ids >= 0, vals[np.maximum(var_ids, 0).astype(np.int64)], u64(0))
    # --- setup tables and sigma
    tables = np.zeros((LOOKUP_WIDTH + 1, n), dtype=np.uint64)
    for ti, t in enumerate(tabs):
        tables[:LOOKUP_WIDTH, offs[ti]: offs[ti] + t.shape[0]] = t.T
        tables[LOOKUP_WIDTH, offs[ti]: offs[ti] + t.shape[0]] = ti + 1
    ks = non_residues(V, n)
    om = F.powers(F.omega(log_n), n)
    sigmas = np.empty((V, n), dtype=np.uint64)
    for c in range(V):
        sigmas[c] = F.mul(om, np.uint64(ks[c]))
    sigma_from_placement(var_ids, next_id, sigmas)
    circuit = Circuit(log_n, GP_COLUMNS, LOOKUP_WIDTH * LOOKUP_REPS, LOOKUP_WIDTH, LOOKUP_REPS, gates, Kc, consts_for_gates,
                      table_id_col, q, variables, mult.reshape(1, n), sigmas, constants, tables, ks, [], total_len,
                      selector_tree=getattr(place_selectors, "last_tree", None), geometry_constant_cols=GEOMETRY_CONSTANTS)
    if not return_info:
        return circuit
    info = dict(digest=bytes(int(v) for v in vals[np.array(output, dtype=np.int64)]), gp_rows=gp_rows, lookup_rows=lk_rows,
                num_variables=int(next_id), num_blocks=(len(message) + 9 + 63) // 64,
                gate_instances={"%s%s" % (k, list(c)): int(g_totals[i]) for i, (k, c) in enumerate(cs.key_list)},
                lookups={t: int(l_totals[t]) for t in sorted(l_totals)},
                var_ids=var_ids, all_values=vals)             # placement and values by variable (WitnessVec + copy hint)
    return circuit, info


def sigma_from_placement(var_ids, num_vars, sigmas):
    """In place: sigmas holds the identities k_col * omega^row on entry (setup.rs:419-503)."""
    V, n = var_ids.shape
    if F._NATIVE is not None:
        var_ids = np.ascontiguousarray(var_ids, dtype=np.int32)
        assert sigmas.flags["C_CONTIGUOUS"]
        F._NATIVE.synth_sigma_from_placement(var_ids.ctypes.data, V, n, int(num_vars), sigmas.ctypes.data)
        return
    flat = var_ids.reshape(-1)
    cells = np.flatnonzero(flat >= 0)
    order = cells[np.argsort(flat[cells], kind="stable")]          # grouped by variable, cells ascending inside
    v = flat[order]
    start = np.flatnonzero(np.concatenate([[True], v[1:] != v[:-1]]))
    end = np.concatenate([start[1:], [len(order)]]) - 1
    prev = np.empty_like(order)
    prev[1:] = order[:-1]
    prev[start] = order[end]
    ident = sigmas.reshape(-1).copy()
    sigmas.reshape(-1)[order] = ident[prev]


def message_len_for_log_n(log_n):
    """Largest whole-block message length (bytes) whose circuit still fits 2^log_n rows.  Instance counts are affine in
    the number of full blocks, so two small syntheses give the exact counts for any length."""
    (_, a), (_, b) = (sha256_circuit(bytes(64 * m), return_info=True) for m in (4, 5))

    def rows(m):
        gp = sum(-(-(a["gate_instances"][k] + (m - 4) * (b["gate_instances"][k] - a["gate_instances"][k])) //
                   CAP[k.split("[")[0]]) for k in a["gate_instances"])
        lk = sum(-(-(a["lookups"][t] + (m - 4) * (b["lookups"][t] - a["lookups"][t])) // LOOKUP_REPS) for t in a["lookups"])
        return max(gp + 1, lk)

    lo, hi = 1, 1 << 24
    assert rows(4) <= (1 << log_n), "2^%d rows cannot hold the tables and four blocks" % log_n
    lo = 4
    while lo < hi:
        mid = (lo + hi + 1) // 2
        if rows(mid) <= (1 << log_n):
            lo = mid
        else:
            hi = mid - 1
    return 64 * lo


def bench_message(n_bytes, seed=42):
    """The bench hashes seeded random bytes (sha256/mod.rs:319-325; the reference's ChaCha stream is not reproduced)."""
    return np.random.default_rng(seed).integers(0, 256, size=n_bytes, dtype=np.uint8).tobytes()
