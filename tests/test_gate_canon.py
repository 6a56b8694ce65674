"""Seam S3 as the reference emits it (src/gpu_synthesizer/mod.rs:210-444): captures of EVERY evaluator in the reference's own
call order and numbering (tests/reference_capture.py) are functions equal to independent formulas, are recognised by the
library whatever their numbering and relation order (csrc/gate_canon.h: structural fingerprint), fit the interpreter's slot
budget by live range, and a host's own gate gets a kernel compiled at run time (csrc/gate_jit.hip; the compilation itself is
checked here without a GPU).  Host-only: runs in the CPU suite."""
import ctypes as C
import random

import pytest

import era_boojum_amd as E
from era_boojum_amd import gate_program as G
from oracle import gates as OG
import reference_capture as RC

P = G.P


class _Lib:      # resolved at first use: the GPU tests import this module and must get torch's HIP runtime loaded first (gpu_util.ctx)
    def __getattr__(self, name):
        if not name.startswith("bj_"):      # pytest probes module-level objects for its own attributes while collecting
            raise AttributeError(name)
        return getattr(E.load_library(), name)


lib = _Lib()


def info(prog):
    fp, ns, no, ext = (C.c_uint64 * 2)(), C.c_uint32(), C.c_uint32(), (C.c_uint32 * 3)()
    rc = lib.bj_gate_program_canonical_info(C.byref(prog.struct), fp, C.byref(ns), C.byref(no), ext)
    assert rc == 0, rc
    return (fp[0], fp[1]), ns.value, no.value, tuple(ext)


def generated(prog):
    return lib.bj_gate_program_generated(C.byref(prog.struct))


e = lambda x: (x % P, 0)
# independent formulas: oracle/gates.py (pinned by the reference's own proof) where the golden circuit has the gate
GOLDEN = {"fma": OG.ev_fma, "fma_product_body": OG.ev_fma, "zero_check": OG.ev_zero_check, "uintx_add": OG.ev_uintx_add,
          "reduction4": OG.ev_reduction4, "constants_allocator": OG.ev_constants_allocator, "boolean": OG.ev_boolean,
          "selection": OG.ev_selection, "parallel_selection4": OG.ev_parallel_selection4, "dot_product4": OG.ev_dot_product4,
          "u8x4_fma": OG.ev_u8x4_fma, "poseidon2_flattened": OG.ev_poseidon2_flattened}


def _ext_mul(x, y):
    return ((x[0] * y[0] + 7 * x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)


def _f_cswap(n):
    def f(v, c, w):
        s, out = v[0], []
        for i in range(n):
            a, b, ra, rb = v[4 * i + 1: 4 * i + 5]
            out += [(b * s + (1 - s) * a - ra) % P, (a * s + (1 - s) * b - rb) % P]
        return out
    return f


def _f_fma_ext(v, c, w):
    a, b, cc, d, q, l = (v[0], v[1]), (v[2], v[3]), (v[4], v[5]), (v[6], v[7]), (c[0], c[1]), (c[2], c[3])
    t, u = _ext_mul(_ext_mul(a, b), q), _ext_mul(cc, l)
    return [(t[0] + u[0] - d[0]) % P, (t[1] + u[1] - d[1]) % P]


def _f_matrix(m):
    return lambda v, c, w: [(sum(m[r][k] * v[k] for k in range(12)) - v[12 + r]) % P for r in range(12)]


FORMULAS = {   # the remaining evaluators: formulas restated from the gates' doc comments / constraint descriptions
    "zero_check_witness_inversion": lambda v, c, w: [(v[1] + v[0] * w[0] - 1) % P, v[0] * v[1] % P],
    "conditional_swap1": _f_cswap(1), "conditional_swap2": _f_cswap(2),
    "quadratic_combination4": lambda v, c, w: [sum(v[2 * i] * v[2 * i + 1] for i in range(4)) % P],
    "reduction_by_powers4": lambda v, c, w: [(sum(v[i] * pow(c[0], i, P) for i in range(4)) - v[4]) % P],
    "simple_non_linearity7": lambda v, c, w: [(pow(v[0] + c[0], 7, P) - v[1]) % P],
    "simple_non_linearity5": lambda v, c, w: [(pow(v[0] + c[0], 5, P) - v[1]) % P],
    "simple_non_linearity3": lambda v, c, w: [(pow(v[0] + c[0], 3, P) - v[1]) % P],
    "u32_add": lambda v, c, w: [(v[0] + v[1] + v[2] - v[3] - (1 << 32) * v[4]) % P, (v[4] * v[4] - v[4]) % P],
    "u32_sub": lambda v, c, w: [(v[0] - v[1] - v[2] - v[3] + (1 << 32) * v[4]) % P, (v[4] * v[4] - v[4]) % P],
    "u32_tri_add_carry_as_chunk": lambda v, c, w: [(sum(v[4 * o + k] << (8 * k) for o in range(3) for k in range(4))
                                                     - sum(v[12 + k] << (8 * k) for k in range(4)) - (v[16] << 32)) % P],
    "fma_in_extension": _f_fma_ext,
    "matrix_multiplication_poseidon2_external": _f_matrix(G.poseidon2_external_matrix()),
    "matrix_multiplication_poseidon2_inner": _f_matrix(G.poseidon2_inner_matrix()),
    "matrix_multiplication_host_matrix": _f_matrix(RC.MATRIX),
}
CAPTURES = RC.all_captures()


def want_terms(name, var, con, wit):
    if name in GOLDEN:
        return [t[0] for t in GOLDEN[name]([e(v) for v in var], [e(c) for c in con])]
    return FORMULAS[name](var, con, wit)


def test_every_evaluator_is_covered():
    assert set(CAPTURES) == set(GOLDEN) | set(FORMULAS)
    assert len(CAPTURES) >= 21 + 1          # the 20 evaluators with terms (+ parametrisations) and the Poseidon2 flattened gate


@pytest.mark.parametrize("name", sorted(CAPTURES))
def test_capture_in_the_references_order_and_numbering(name):
    thunk, nv, nc, nw = CAPTURES[name]
    cap = thunk()
    tmps = [dst[1] for dst, _ in cap.relations]
    assert min(tmps) > 900 and len(set(tmps)) == len(tmps) and tmps == sorted(tmps)     # process-wide counter: fresh, never reused
    dense, raw = RC.to_program(cap), RC.to_program_raw(cap)
    assert dense.num_temporaries == len(cap.relations) and raw.num_temporaries > max(tmps)
    rnd = random.Random(hash(name) & 0xFFFF)
    for _ in range(3 if len(cap.relations) > 1000 else 20):
        var, con, wit = ([rnd.randrange(P) for _ in range(k)] for k in (nv, nc, nw))
        want = want_terms(name, var, con, wit)
        assert dense.evaluate(var, con, wit) == want
        assert raw.evaluate(var, con, wit) == want
    # one function, one fingerprint: the reference's sparse numbers, the Rust shim's dense ones, a second capture made later
    # in the same process (other numbers again) — and the library has a build-time kernel for it unless the host chose it
    fp = info(dense)[0]
    assert info(raw)[0] == fp and info(RC.to_program_raw(thunk()))[0] == fp
    fp_, slots, ops, ext = info(dense)
    assert ext == (nv, nc, nw)
    assert slots <= 64, (name, slots)                       # 288 relations of a matrix gate, ~9.6 k of Poseidon2: all fit
    assert generated(dense) == generated(raw) == (0 if name == "matrix_multiplication_host_matrix" else 1), name


def _shuffle_independent(prog, rnd):
    """The same DAG with its relations in another valid order and its temporaries renumbered at random."""
    rel = list(prog.relations)
    n = len(rel)
    defined_at = {dst: i for i, (op, dst, a, b) in enumerate(rel)}          # SSA input: one definition per temporary
    deps = []
    for op, dst, a, b in rel:
        d = set()
        for k, ix in ((a, b) if op in (G.OP_ADD, G.OP_SUB, G.OP_MUL) else (a,)):
            if k == G.IDX_TEMPORARY:
                d.add(defined_at[ix])
        deps.append(d)
    done, order, ready = set(), [], [i for i in range(n) if not deps[i]]
    while ready:
        i = ready.pop(rnd.randrange(len(ready)))
        order.append(i)
        done.add(i)
        for j in range(n):
            if j not in done and j not in ready and deps[j] <= done:
                ready.append(j)
    names = list(range(3 * n))
    rnd.shuffle(names)
    ren = {rel[i][1]: names[k] for k, i in enumerate(order)}
    fix = lambda ref: (ref[0], ren[ref[1]]) if ref[0] == G.IDX_TEMPORARY else ref
    new_rel = [(rel[i][0], ren[rel[i][1]], fix(rel[i][2]), fix(rel[i][3]) if rel[i][0] in (G.OP_ADD, G.OP_SUB, G.OP_MUL) else (0, 0))
               for i in order]
    return G.GateProgram(new_rel, prog.values, [fix(w) for w in prog.writes], 3 * n)


@pytest.mark.parametrize("name", ["fma", "u8x4_fma", "fma_in_extension", "parallel_selection4", "matrix_multiplication_host_matrix",
                                  "u32_tri_add_carry_as_chunk"])
def test_fingerprint_is_invariant_under_reordering_and_renumbering(name):
    thunk, nv, nc, nw = CAPTURES[name]
    prog = RC.to_program(thunk())
    fp = info(prog)[0]
    rnd = random.Random(len(name))
    for _ in range(5):
        other = _shuffle_independent(prog, rnd)
        assert [r[1] for r in other.relations] != [r[1] for r in prog.relations]
        var, con, wit = ([rnd.randrange(P) for _ in range(k)] for k in (nv, nc, nw))
        assert other.evaluate(var, con, wit) == prog.evaluate(var, con, wit)
        assert info(other)[0] == fp
    # commutative operands swapped, x + 0 / x * 1 padding: still the same function, same fingerprint
    rel = [(op, dst, b, a) if op in (G.OP_ADD, G.OP_MUL) else (op, dst, a, b) for op, dst, a, b in prog.relations]
    assert info(G.GateProgram(rel, prog.values, prog.writes, prog.num_temporaries))[0] == fp
    # ... and a different function has a different one
    rel = list(prog.relations)
    k = next(i for i, r in enumerate(rel) if r[0] == G.OP_MUL)
    rel[k] = (G.OP_ADD,) + rel[k][1:]
    assert info(G.GateProgram(rel, prog.values, prog.writes, prog.num_temporaries))[0] != fp


def test_the_tracer_programs_are_the_captures():
    """era_boojum_amd/gate_program.py's programs — what gate_codegen.py generates kernels from — and the captures made under the
    reference's numbering have the same fingerprints, the independently transliterated ones included."""
    pairs = {"fma": G.fma_program(), "zero_check": G.zero_check_program(), "uintx_add": G.uintx_add_program(),
             "zero_check_witness_inversion": G.zero_check_program(True), "u8x4_fma": G.u8x4_fma_program(),
             "poseidon2_flattened": G.poseidon2_flattened_program()}
    for name, prog in pairs.items():
        assert info(prog)[0] == info(RC.to_program_raw(CAPTURES[name][0]()))[0], name


def test_malformed_lists_are_refused():
    ok = G.fma_program()
    bad_use = G.GateProgram([(G.OP_ADD, 0, (G.IDX_TEMPORARY, 1), (G.IDX_VARIABLE, 0))], [], [(G.IDX_TEMPORARY, 0)], 2)   # reads t1 before any write
    bad_val = G.GateProgram([(G.OP_ADD, 0, (G.IDX_VALUE, 3), (G.IDX_VARIABLE, 0))], [1], [(G.IDX_TEMPORARY, 0)], 1)
    bad_dst = G.GateProgram([(G.OP_ADD, 5, (G.IDX_VARIABLE, 1), (G.IDX_VARIABLE, 0))], [], [(G.IDX_TEMPORARY, 0)], 1)
    for prog, rc in ((ok, 0), (bad_use, -1), (bad_val, -1), (bad_dst, -1)):
        assert lib.bj_gate_program_canonical_info(C.byref(prog.struct), None, None, None, None) == rc


def _source(prog):
    need = lib.bj_gate_program_jit_source(C.byref(prog.struct), None, 0)
    buf = C.create_string_buffer(need)
    lib.bj_gate_program_jit_source(C.byref(prog.struct), buf, need)
    return buf.value.decode()


def test_a_hosts_own_gate_compiles_at_run_time():
    """What bj_setup_create does for a program without a build-time kernel, up to loading the code object: the HIP source the
    library writes for the canonical program compiles with hiprtc for gfx950 (no GPU needed for that)."""
    b = G.GateProgramBuilder()
    x, y, z = b.var(0), b.var(1), b.var(2)
    b.push((x * y - z).square() * b.const_poly(0) + 5)
    b.push((x + y).inverse() - z)
    own = b.build()
    assert generated(own) == 0
    src = _source(own)
    assert "bj_jit_gate" in src and "inv_pow" in src and "term[1]" in src
    log = C.create_string_buffer(4096)
    for prog in (own, RC.to_program(CAPTURES["matrix_multiplication_host_matrix"][0]())):
        rc = lib.bj_gate_program_jit_compile_check(C.byref(prog.struct), b"gfx950", log, 4096)
        if rc == -5:
            pytest.skip("hiprtc is not installed here: " + log.value.decode())
        assert rc == 0, log.value.decode()


def test_random_malformed_lists_never_crash_the_canonicaliser():
    """Random relations with out-of-range kinds, operations, temporaries, value indices and column numbers: every list is either
    accepted (and then has a body) or refused with BJ_ERR_INVALID_ARG — bj_setup_create validates a host's op lists with this."""
    rnd = random.Random(1)
    accepted = 0
    for _ in range(4000):
        nrel, ntmp, nval = rnd.randrange(0, 12), rnd.randrange(0, 8), rnd.randrange(0, 3)
        ref = lambda: (rnd.choice([0, 1, 2, 3, 4, 5, 9]), rnd.choice([0, 1, 2, 5, 1 << 19, 1 << 20, (1 << 32) - 1]))
        rel = [(rnd.choice([0, 1, 2, 3, 4, 5, 6, 7, 8, 99]), rnd.randrange(0, 10), ref(), ref()) for _ in range(nrel)]
        prog = G.GateProgram(rel, [rnd.randrange(1 << 64) for _ in range(nval)], [ref() for _ in range(rnd.randrange(0, 4))], ntmp)
        prog.struct.num_values = nval
        rc = lib.bj_gate_program_canonical_info(C.byref(prog.struct), None, None, None, None)
        assert rc in (0, -1)
        assert (rc == 0) == (lib.bj_gate_program_emit_body(C.byref(prog.struct), None, 0) > 0)
        accepted += rc == 0
    assert 0 < accepted < 4000
