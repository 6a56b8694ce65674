"""Gate op lists (era_boojum_amd/gate_program.py, seam S3) for the evaluators that the golden proof's circuit does not
contain: each traced program against an independent restatement of the reference formula in python integers (file:line in
the program's docstring), on random inputs and on a satisfying assignment; plus the builder's slot renaming."""
import random

import pytest

from era_boojum_amd import gate_program as G

P = G.P
rnd = random.Random(20240807)
rv = lambda k: [rnd.randrange(P) for _ in range(k)]
MATRIX = [[rnd.randrange(1, 1 << 16) for _ in range(12)] for _ in range(12)]


def f_conditional_swap(v, c):
    s, out = v[0], []
    for i in range(2):
        a, b, ra, rb = v[4 * i + 1: 4 * i + 5]
        out += [(b * s + (1 - s) * a - ra) % P, (a * s + (1 - s) * b - rb) % P]
    return out


def f_fma_ext(v, c):
    mul = lambda x, y: ((x[0] * y[0] + 7 * x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)
    a, b, cc, d = (v[0], v[1]), (v[2], v[3]), (v[4], v[5]), (v[6], v[7])
    q, l = (c[0], c[1]), (c[2], c[3])
    t, u = mul(mul(a, b), q), mul(cc, l)
    return [(t[0] + u[0] - d[0]) % P, (t[1] + u[1] - d[1]) % P]


CASES = [  # program, formula, variables, constants, a satisfying assignment (or None)
    (G.conditional_swap_program(2), f_conditional_swap, 9, 0,
     lambda: (lambda a, b, c, d: [1, a, b, b, a, c, d, d, c])(*rv(4))),
    (G.quadratic_combination_program(4), lambda v, c: [sum(v[2 * i] * v[2 * i + 1] for i in range(4)) % P], 8, 0,
     lambda: (lambda a, b: [a, b, P - a, b, 0, 5, 7, 0])(*rv(2))),
    (G.reduction_by_powers_program(4), lambda v, c: [(sum(v[i] * pow(c[0], i, P) for i in range(4)) - v[4]) % P], 5, 1, None),
    (G.simple_non_linearity_program(7), lambda v, c: [(pow(v[0] + c[0], 7, P) - v[1]) % P], 2, 1, None),
    (G.simple_non_linearity_program(5), lambda v, c: [(pow(v[0] + c[0], 5, P) - v[1]) % P], 2, 1, None),
    (G.u32_add_program(), lambda v, c: [(v[0] + v[1] + v[2] - v[3] - (1 << 32) * v[4]) % P, (v[4] * v[4] - v[4]) % P], 5, 0,
     lambda: [0xFFFFFFFF, 5, 1, 5, 1]),
    (G.u32_sub_program(), lambda v, c: [(v[0] - v[1] - v[2] - v[3] + (1 << 32) * v[4]) % P, (v[4] * v[4] - v[4]) % P], 5, 0,
     lambda: [3, 5, 1, 0xFFFFFFFD, 1]),
    (G.u32_tri_add_carry_as_chunk_program(),
     lambda v, c: [(sum(v[4 * o + k] << (8 * k) for o in range(3) for k in range(4)) - sum(v[12 + k] << (8 * k) for k in range(4))
                    - (v[16] << 32)) % P], 17, 0,
     lambda: (lambda x, y, z: [*x.to_bytes(4, "little"), *y.to_bytes(4, "little"), *z.to_bytes(4, "little"),
                               *((x + y + z) & 0xFFFFFFFF).to_bytes(4, "little"), (x + y + z) >> 32])(0xFFFFFFF0, 0xDEADBEEF, 0x12345678)),
    (G.fma_in_extension_program(), f_fma_ext, 8, 4, None),
    (G.matrix_multiplication_program(MATRIX),
     lambda v, c: [(sum(MATRIX[r][k] * v[k] for k in range(12)) - v[12 + r]) % P for r in range(12)], 24, 0,
     lambda: (lambda x: x + [sum(MATRIX[r][k] * x[k] for k in range(12)) % P for r in range(12)])(rv(12))),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_traced_program_equals_the_reference_formula(case):
    prog, formula, nv, nc, sat = CASES[case]
    for _ in range(20):
        v, c = rv(nv), rv(nc)
        assert prog.evaluate(v, c) == formula(v, c)
    if sat is not None:
        v = [x % P for x in sat()]
        assert prog.evaluate(v, rv(nc)) == [0] * prog.num_terms, "a satisfying assignment must zero every term"
    if nc and sat is None:                                  # solve for the result variable: satisfiable by construction
        v, c = rv(nv), rv(nc)
        t = formula(v, c)
        if prog.num_terms == 1:
            v[-1] = (v[-1] + t[0]) % P
            assert prog.evaluate(v, c) == [0]


def _canonical_info(prog):
    import ctypes as C
    import era_boojum_amd as E
    lib = E.load_library()
    fp, ns, no, ext = (C.c_uint64 * 2)(), C.c_uint32(), C.c_uint32(), (C.c_uint32 * 3)()
    rc = lib.bj_gate_program_canonical_info(C.byref(prog.struct), fp, C.byref(ns), C.byref(no), ext)
    assert rc == 0, rc
    return (fp[0], fp[1]), ns.value, no.value, tuple(ext)


def test_the_library_assigns_slots_by_live_range_whatever_the_numbering():
    b = G.GateProgramBuilder()
    xs = [b.var(i) for i in range(6)]
    acc = xs[0] * xs[1]
    for i in range(200):                                   # a long chain: every intermediate dies at once
        acc = acc * xs[i % 6] + xs[(i + 1) % 6]
    keep = xs[2] * xs[3]                                   # a term computed early and written late
    b.push(acc)
    b.push(keep - acc)
    prog = b.build()
    assert len(prog.relations) == 403 and prog.num_temporaries == 403        # one temporary per relation, like the reference
    fp, slots, ops, ext = _canonical_info(prog)
    assert slots <= 3 and ops == 403 and ext == (6, 0, 0)
    v = rv(6)
    acc = v[0] * v[1]
    for i in range(200):
        acc = (acc * v[i % 6] + v[(i + 1) % 6]) % P
    assert prog.evaluate(v, []) == [acc % P, (v[2] * v[3] - acc) % P]


def test_poseidon2_flattened_gate_equals_the_golden_pinned_evaluator():
    """The 118-relation gate of the recursion circuits (poseidon2.rs:165-410) as a traced op list against
    oracle/gates.py::ev_poseidon2_flattened, whose formulas the reference's own proof pins (tests/test_oracle_fixture.py)."""
    from oracle import gates as OG
    prog, compact = G.poseidon2_flattened_program(), G.poseidon2_flattened_compact_program()
    assert prog.num_terms == 118 and 9000 < len(prog.relations) < 10000       # dense 12 x 12 layers, as the reference records them
    assert compact.num_terms == 118 and len(compact.relations) < 3000
    for _ in range(3):
        v = rv(130)
        want = [t[0] for t in OG.ev_poseidon2_flattened([(x, 0) for x in v], [])]
        assert prog.evaluate(v, []) == want and compact.evaluate(v, []) == want
    fp, slots, ops, ext = _canonical_info(prog)
    assert slots <= 64 and ops < 6000 and ext == (130, 0, 0)                  # x*1 and 0+x gone, shared prefix sums merged


def test_oracle_prover_handles_op_list_and_specialized_gates():
    """The CPU oracle prover on a circuit of the golden proof's class (op-list gates, the Poseidon2 flattened gate, a
    boolean specialized column): its proof is accepted by the verifier restatement, a broken Poseidon2 row is caught."""
    import numpy as np
    from era_boojum_amd import synthetic as S
    from oracle import prover as OP, verifier as OV
    c = S.recursion_like_circuit(8, seed=3)
    st = OP.Setup(c, 2, 8, threads=4)
    assert OV.verify(OV.VerificationKey(c, st.cap, 2, 8), OP.prove(c, st, 2, 8, security_level=20, threads=4))
    g = c.gates[2]
    m = np.ones(c.n, dtype=bool)
    for i, bit in enumerate(g.path):
        m &= c.constants[i] == (1 if bit else 0)
    row = int(np.flatnonzero(m)[0])
    c.variables[50, row] = (int(c.variables[50, row]) + 1) % P
    with pytest.raises(AssertionError, match="unsatisfied"):
        OP.prove(c, st, 2, 8, security_level=20, threads=4)


def test_generated_kernels_cover_the_known_programs_and_the_committed_file_is_current():
    """csrc/gate_aot.hip is what gate_codegen.generate() emits today, and the library recognises the known programs by the
    structural fingerprint of their canonical form (csrc/gate_canon.cpp: the generator asks the same code)."""
    import ctypes as C
    import os
    import era_boojum_amd as E
    from era_boojum_amd import gate_codegen as GC
    path = os.path.join(os.path.dirname(os.path.abspath(GC.__file__)), "csrc", "gate_aot.hip")
    assert open(path).read() == GC.generate(), "run `python -m era_boojum_amd.gate_codegen` and rebuild"
    lib = E.load_library()
    for name, prog in GC.known_programs().items():
        assert lib.bj_gate_program_generated(C.byref(prog.struct)) == 1, name
        assert _canonical_info(prog)[0] == GC.program_fingerprint(prog)       # helper library == product library
    assert lib.bj_gate_program_generated(C.byref(G.matrix_multiplication_program(MATRIX).struct)) == 0     # host-chosen matrix: compiled at run time
    assert lib.bj_gate_program_generated(C.byref(G.poseidon2_flattened_program().struct)) == 1            # -> the hand-written evaluator
    assert lib.bj_gate_program_generated(C.byref(G.poseidon2_flattened_program(12).struct)) == 0          # other cell placement: compiled
    b = G.GateProgramBuilder()
    b.push(b.var(0) * b.var(1) - b.var(2) + 5)
    assert lib.bj_gate_program_generated(C.byref(b.build().struct)) == 0                                   # a host's own gate


