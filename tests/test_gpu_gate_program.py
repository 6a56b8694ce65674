"""-m gpu: seam S3 — gates given as op lists (bj_gate_program, what the reference's gpu_synthesizer captures) evaluated
by the interpreter kernel: raw terms against the golden-pinned evaluators of oracle/gates.py, and a whole proof whose
gates all go through op lists against the oracle prover's proof."""
import numpy as np
import pytest

import era_boojum_amd as E
from era_boojum_amd import gate_program as GP, proof_format, synthetic as S
from gpu_util import DevBuf, ctx, rand_gl, P
from oracle import gates as OG
from oracle import prover as OP
from oracle import verifier as OV

pytestmark = pytest.mark.gpu

CASES = [  # program, oracle evaluator, principal width, row constants, repetitions, per-repetition constant stride
    (GP.fma_program, "FmaGateInBaseFieldWithoutConstant", 4, 2, 3, 0),
    (GP.reduction4_program, "ReductionGate<4>", 5, 4, 2, 0),
    (GP.constants_allocator_program, "ConstantsAllocatorGate", 1, 1, 4, 1),
    (GP.selection_program, "SelectionGate", 4, 0, 3, 0),
    (GP.dot_product4_program, "DotProductGate<4>", 9, 0, 2, 0),
    (GP.zero_check_program, "ZeroCheckGate", 3, 0, 3, 0),
    (GP.uintx_add_program, "UIntXAddGate", 5, 1, 2, 0),
    (GP.boolean_program, "BooleanConstraintGate", 1, 0, 5, 0),
    (GP.parallel_selection4_program, "ParallelSelectionGate<4>", 13, 0, 2, 0),
    (GP.u8x4_fma_program, "U8x4FMAGate", 26, 0, 2, 0),
]


@pytest.mark.parametrize("make,name,width,n_const,reps,cstride", CASES)
def test_program_terms_match_oracle_evaluators(make, name, width, n_const, reps, cstride):
    prog = make()
    n_points = 1000
    rng = np.random.default_rng(len(name))
    n_con_cols = max(1, n_const + (reps - 1) * cstride)
    var = rand_gl(rng, (width * reps, n_points), noncanonical=True)
    con = rand_gl(rng, (n_con_cols, n_points), noncanonical=True)
    d_var, d_con = DevBuf(var), DevBuf(con)
    d_out = DevBuf(nelems=reps * prog.num_terms * n_points)
    ctx().gate_program_eval(prog, d_var.ptr, n_points, d_con.ptr, n_points, reps, width, cstride, n_points, d_out.ptr)
    got = d_out.get((reps, prog.num_terms, n_points))
    fn = OG.EVALUATORS[name][5]
    for i in (0, 1, 17, n_points - 1):
        for r in range(reps):
            v = [(int(x) % P, 0) for x in var[r * width:(r + 1) * width, i]]
            c = [(int(x) % P, 0) for x in con[r * cstride:, i]]
            want = [t[0] for t in fn(v, c)]
            assert [int(x) for x in got[r, :, i]] == want, (name, i, r)
    # all points against the program's own python semantics (vectorised over a few hundred points would be slow: sample)
    for i in range(0, n_points, 97):
        for r in range(reps):
            want = prog.evaluate([int(x) for x in var[r * width:(r + 1) * width, i]], [int(x) for x in con[r * cstride:, i]])
            assert [int(x) for x in got[r, :, i]] == want


def test_proof_with_op_list_gates_equals_oracle_proof():
    """Every gate of the SHA-shaped circuit handed over as an op list (kind BJ_GATE_PROGRAM): same proof, bit for bit."""
    c = S.sha_shaped_circuit(10, seed=31, table_bits=2)
    osetup = OP.Setup(c, 8, 16, threads=4)
    po = OP.prove(c, osetup, 8, 16, security_level=30, threads=4)
    progs = {"ConstantsAllocatorGate": GP.constants_allocator_program(), "FmaGateInBaseFieldWithoutConstant": GP.fma_program(),
             "ReductionGate<4>": GP.reduction4_program()}
    for g in c.gates:
        if g.name in progs:
            g.program = progs[g.name]
    gsetup = E.ProverSetup(ctx(), c, 8, 16, 30)
    buf, _ = gsetup.prove()
    pg = proof_format.parse(buf, security_level=30)
    for k in ("witness_oracle_cap", "stage_2_oracle_cap", "quotient_oracle_cap", "values_at_z", "values_at_z_omega", "values_at_0",
              "fri_base_oracle_cap", "fri_intermediate_oracles_caps", "final_fri_monomials", "queries_per_fri_repetition"):
        assert pg[k] == po[k], k
    # mixed: one gate hand-written, the others from op lists
    for g in c.gates:
        if g.name == "FmaGateInBaseFieldWithoutConstant":
            g.program = None
    gsetup2 = E.ProverSetup(ctx(), c, 8, 16, 30)
    buf2, _ = gsetup2.prove()
    assert np.array_equal(buf, buf2)
    gsetup.close(); gsetup2.close()


def test_bad_programs_are_rejected():
    d = DevBuf(nelems=64)
    reads_unwritten = GP.GateProgram([(GP.OP_ADD, 0, (GP.IDX_TEMPORARY, 1), (GP.IDX_VARIABLE, 0))], [], [(GP.IDX_TEMPORARY, 0)], 2)
    dst_out_of_range = GP.GateProgram([(GP.OP_ADD, 7, (GP.IDX_VARIABLE, 1), (GP.IDX_VARIABLE, 0))], [], [(GP.IDX_TEMPORARY, 0)], 1)
    for prog in (reads_unwritten, dst_out_of_range):
        with pytest.raises(E.BoojumHipError):
            ctx().gate_program_eval(prog, d.ptr, 8, d.ptr, 8, 1, 2, 0, 8, d.ptr)
    b = GP.GateProgramBuilder()                             # any number of temporaries is fine: slots go by live range
    acc = b.var(0) * b.var(1)
    for _ in range(2000):
        acc = acc * b.var(0) + b.var(1)
    b.push(acc)
    prog = b.build()
    assert prog.num_temporaries == 4001
    ctx().gate_program_eval(prog, d.ptr, 8, d.ptr, 8, 1, 2, 0, 8, d.ptr + 8 * 16)


def test_circuit_with_more_gate_types_proves_and_verifies():
    """Seven evaluator types, three of them (Selection, ZeroCheck with two terms, UIntXAdd with two terms and a row-shared
    constant) known to the prover ONLY as op lists; quotient degree 8.  No oracle prover for this gate set, so the proof is
    checked by the verifier restatement (Merkle paths, DEEP, FRI, quotient identity with the golden-pinned evaluators) and
    by the golden-pinned identity code on the emitted VerificationKey JSON; a broken witness must be reported."""
    import json
    import oracle as O
    from oracle import golden_quotient as GQ
    from oracle import verifier as OV
    from era_boojum_amd import wire_format as W
    c = S.sha_shaped_circuit(11, seed=77, table_bits=2, extended=True)
    assert c.quotient_degree == 8 and [g.name for g in c.gates][3:6] == ["SelectionGate", "ZeroCheckGate", "UIntXAddGate"]
    gsetup = E.ProverSetup(ctx(), c, 8, 16, 40)
    buf, _ = gsetup.prove()
    pg = proof_format.parse(buf, security_level=40)
    assert OV.verify(OV.VerificationKey(c, gsetup.cap(), 8, 16), pg, verbose=True)
    vk = json.loads(W.dumps(W.vk_to_reference_json(c, gsetup.cap(), 8, 16)))
    t = O.Transcript()
    t.absorb_cap(gsetup.cap())
    t.absorb(pg["public_inputs"])
    t.absorb_cap(np.array(pg["witness_oracle_cap"], dtype=np.uint64))
    beta, gamma, lbeta, lgamma = (t.challenge_ext() for _ in range(4))
    t.absorb_cap(np.array(pg["stage_2_oracle_cap"], dtype=np.uint64))
    alpha = t.challenge_ext()
    t.absorb_cap(np.array(pg["quotient_oracle_cap"], dtype=np.uint64))
    z = t.challenge_ext()
    lhs, rhs = GQ.quotient_identity(GQ.geometry_from_vk_json(vk), [g.name for g in c.gates], [], c.non_residues,
                                    dict(beta=beta, gamma=gamma, lookup_beta=lbeta, lookup_gamma=lgamma, alpha=alpha, z=z),
                                    pg["values_at_z"], pg["values_at_z_omega"][0])
    assert lhs == rhs
    # break one ZeroCheck row: input * flag != 0
    zc = next(g for g in c.gates if g.name == "ZeroCheckGate")
    m = np.ones(c.n, dtype=bool)
    for i, bit in enumerate(zc.path):
        m &= c.constants[i] == (1 if bit else 0)
    row = int(np.nonzero(m)[0][0])
    bad = c.variables.copy()
    bad[1, row] = 1                      # flag := 1
    bad[0, row] = 5                      # input := 5   -> input * flag = 5
    with pytest.raises(E.BoojumHipError, match="not satisfied"):
        gsetup.prove(variables=bad)
    gsetup.close()


MORE = [  # evaluators the golden proof's circuit does not contain: program, principal width, row constants, repetitions
    (GP.conditional_swap_program(2), 9, 0, 2),
    (GP.quadratic_combination_program(4), 8, 0, 3),
    (GP.reduction_by_powers_program(4), 5, 1, 2),
    (GP.simple_non_linearity_program(7), 2, 1, 4),
    (GP.u32_add_program(), 5, 0, 3),
    (GP.u32_sub_program(), 5, 0, 3),
    (GP.u32_tri_add_carry_as_chunk_program(), 17, 0, 2),
    (GP.fma_in_extension_program(), 8, 4, 2),
    (GP.matrix_multiplication_program([[(7 * r + 3 * c + 1) % 23 + 1 for c in range(12)] for r in range(12)]), 24, 0, 2),
]


@pytest.mark.parametrize("case", range(len(MORE)))
def test_interpreter_runs_the_remaining_evaluators(case):
    """The other gates of src/cs/gates over general-purpose columns (formulas pinned against independent restatements in
    tests/test_gate_programs.py): the interpreter kernel computes what the op list says, with renamed temporary slots."""
    prog, width, n_const, reps = MORE[case]
    n_points = 600
    rng = np.random.default_rng(100 + case)
    var = rand_gl(rng, (width * reps, n_points), noncanonical=True)
    con = rand_gl(rng, (max(1, n_const), n_points), noncanonical=True)
    d_var, d_con = DevBuf(var), DevBuf(con)
    d_out = DevBuf(nelems=reps * prog.num_terms * n_points)
    ctx().gate_program_eval(prog, d_var.ptr, n_points, d_con.ptr, n_points, reps, width, 0, n_points, d_out.ptr)
    got = d_out.get((reps, prog.num_terms, n_points))
    for i in range(0, n_points, 53):
        for r in range(reps):
            want = prog.evaluate([int(x) for x in var[r * width:(r + 1) * width, i]], [int(x) for x in con[:, i]])
            assert [int(x) for x in got[r, :, i]] == want, (case, i, r)


def test_poseidon2_flattened_gate_through_the_interpreter():
    """The largest evaluator of the reference (118 terms over 130 variables) exactly as the reference captures it — dense 12 x 12
    linear layers, 9.6 k recorded relations on as many temporaries — through the op-list interpreter (raw-terms mode; canonical
    form: ~5.5 k operations on < 64 slots) against the golden-pinned oracle evaluator."""
    prog = GP.poseidon2_flattened_program()
    n_points = 300
    rng = np.random.default_rng(5)
    var = rand_gl(rng, (130, n_points), noncanonical=True)
    con = rand_gl(rng, (1, n_points))
    d_var, d_con = DevBuf(var), DevBuf(con)
    d_out = DevBuf(nelems=prog.num_terms * n_points)
    ctx().gate_program_eval(prog, d_var.ptr, n_points, d_con.ptr, n_points, 1, 130, 0, n_points, d_out.ptr)
    got = d_out.get((1, prog.num_terms, n_points))
    for i in (0, 1, 150, n_points - 1):
        want = [t[0] for t in OG.ev_poseidon2_flattened([(int(x) % P, 0) for x in var[:, i]], [])]
        assert [int(x) for x in got[0, :, i]] == want, i


def test_gate_over_specialized_columns():
    """BooleanConstraintGate placed with UseSpecializedColumns (three repetitions over their own columns after the lookup
    ones, no selector, terms between the lookup and the general-purpose terms in the alpha order — the placement the golden
    proof's circuit uses): the HIP proof is accepted by the verifier restatement AND satisfies the golden-pinned quotient
    identity code, which handles specialized gates exactly as it does for the reference's own proof; a non-bit is reported."""
    import json
    import oracle as O
    from oracle import golden_quotient as GQ
    from oracle import verifier as OV
    from era_boojum_amd import wire_format as W
    c = S.sha_shaped_circuit(10, seed=21, table_bits=2, boolean_columns=3)
    assert c.num_vars == 60 + 32 + 3 and S.check_satisfied(c)
    gsetup = E.ProverSetup(ctx(), c, 8, 16, 30)
    buf, _ = gsetup.prove()
    pg = proof_format.parse(buf, security_level=30)
    assert len(pg["values_at_z"]) > 2 * 95
    assert OV.verify(OV.VerificationKey(c, gsetup.cap(), 8, 16), pg, verbose=True)
    vk = json.loads(W.dumps(W.vk_to_reference_json(c, gsetup.cap(), 8, 16)))
    t = O.Transcript()
    t.absorb_cap(gsetup.cap())
    t.absorb(pg["public_inputs"])
    t.absorb_cap(np.array(pg["witness_oracle_cap"], dtype=np.uint64))
    beta, gamma, lbeta, lgamma = (t.challenge_ext() for _ in range(4))
    t.absorb_cap(np.array(pg["stage_2_oracle_cap"], dtype=np.uint64))
    alpha = t.challenge_ext()
    t.absorb_cap(np.array(pg["quotient_oracle_cap"], dtype=np.uint64))
    z = t.challenge_ext()
    lhs, rhs = GQ.quotient_identity(GQ.geometry_from_vk_json(vk), [g.name for g in c.gates], [("BooleanConstraintGate", 3)],
                                    c.non_residues, dict(beta=beta, gamma=gamma, lookup_beta=lbeta, lookup_gamma=lgamma,
                                                         alpha=alpha, z=z), pg["values_at_z"], pg["values_at_z_omega"][0])
    assert lhs == rhs
    # without the specialized gate in the verifier's configuration the same proof is rejected (the alpha order shifts)
    c_no = S.sha_shaped_circuit(10, seed=21, table_bits=2, boolean_columns=3)
    vk_no = OV.VerificationKey(c_no, gsetup.cap(), 8, 16)
    vk_no.specialized_gates = []
    assert not OV.verify(vk_no, pg)
    bad = c.variables.copy()
    bad[c.num_vars - 2, 77] = 2
    with pytest.raises(E.BoojumHipError, match="not satisfied"):
        gsetup.prove(variables=bad)
    gsetup.close()
    # a specialized gate that reads a constant column is refused
    c.specialized_gates[0].program = GP.constants_allocator_program()
    with pytest.raises(E.BoojumHipError, match="constant"):
        E.ProverSetup(ctx(), c, 8, 16, 30)


@pytest.mark.parametrize("fri_lde,cap", [(2, 32), (8, 16)])
def test_circuit_of_the_golden_proofs_class_proves_and_verifies(fri_lde, cap):
    """Geometry and gate set of the golden proof's inner circuit (a recursion-layer circuit): 130 general-purpose + 8 x 3
    lookup + 1 boolean specialized column = 155 variable columns, eleven evaluators over general-purpose columns incl. the
    118-term Poseidon2 flattened gate, quotient degree 8 — with the golden proof's own FRI parameters (LDE factor 2, cap 32)
    and with the bench's.  The proof is accepted by the verifier restatement and satisfies the quotient identity code that
    the reference's own proof pins, fed with the VerificationKey JSON this repository emits."""
    import json
    import oracle as O
    from oracle import golden_quotient as GQ
    from oracle import verifier as OV
    from era_boojum_amd import wire_format as W
    c = S.recursion_like_circuit(10, seed=4)
    assert (c.num_vars, c.quotient_degree, len(c.gates)) == (155, 8, 11) and S.check_satisfied(c)
    gsetup = E.ProverSetup(ctx(), c, fri_lde, cap, 40)
    buf, _ = gsetup.prove()
    pg = proof_format.parse(buf, security_level=40)
    assert OV.verify(OV.VerificationKey(c, gsetup.cap(), fri_lde, cap), pg, verbose=True)
    vk = json.loads(W.dumps(W.vk_to_reference_json(c, gsetup.cap(), fri_lde, cap)))
    t = O.Transcript()
    t.absorb_cap(gsetup.cap())
    t.absorb(pg["public_inputs"])
    t.absorb_cap(np.array(pg["witness_oracle_cap"], dtype=np.uint64))
    beta, gamma, lbeta, lgamma = (t.challenge_ext() for _ in range(4))
    t.absorb_cap(np.array(pg["stage_2_oracle_cap"], dtype=np.uint64))
    alpha = t.challenge_ext()
    t.absorb_cap(np.array(pg["quotient_oracle_cap"], dtype=np.uint64))
    z = t.challenge_ext()
    lhs, rhs = GQ.quotient_identity(GQ.geometry_from_vk_json(vk), [g.name for g in c.gates], [("BooleanConstraintGate", 1)],
                                    c.non_residues, dict(beta=beta, gamma=gamma, lookup_beta=lbeta, lookup_gamma=lgamma,
                                                         alpha=alpha, z=z), pg["values_at_z"], pg["values_at_z_omega"][0])
    assert lhs == rhs
    # one wrong S-box input inside a Poseidon2 row
    g = c.gates[2]
    m = np.ones(c.n, dtype=bool)
    for i, bit in enumerate(g.path):
        m &= c.constants[i] == (1 if bit else 0)
    bad = c.variables.copy()
    row = int(np.flatnonzero(m)[3])
    bad[77, row] = (int(bad[77, row]) + 1) % P
    with pytest.raises(E.BoojumHipError, match="not satisfied"):
        gsetup.prove(variables=bad)
    gsetup.close()


def test_hand_written_poseidon2_gate_gives_the_same_proof_as_its_op_list():
    """BJ_GATE_POSEIDON2_FLATTENED (csrc/gate_poseidon2.hip) against the same gate through the op-list interpreter: the two
    proofs of the same recursion-class circuit are identical bytes."""
    a = S.recursion_like_circuit(10, seed=9)
    b = S.recursion_like_circuit(10, seed=9, poseidon2_as_op_list=True)
    assert [g.kind for g in a.gates][2] == 6 and [g.kind for g in b.gates][2] == 5 and np.array_equal(a.variables, b.variables)
    sa, sb = E.ProverSetup(ctx(), a, 2, 32, 30), E.ProverSetup(ctx(), b, 2, 32, 30)
    pa, _ = sa.prove()
    pb, _ = sb.prove()
    assert np.array_equal(pa, pb)
    sa.close(); sb.close()


@pytest.mark.parametrize("make,fri_lde,cap", [
    (lambda: S.recursion_like_circuit(9, seed=11), 2, 32),                                   # the golden proof's class and parameters
    (lambda: S.recursion_like_circuit(8, seed=12, poseidon2_as_op_list=True), 8, 16),
    (lambda: S.sha_shaped_circuit(9, seed=13, table_bits=2, extended=True), 8, 16),          # op-list gates with two terms / constants
    (lambda: S.sha_shaped_circuit(9, seed=14, table_bits=2, boolean_columns=3), 4, 8),       # gates over specialized columns
])
def test_hip_proof_equals_oracle_proof_with_op_list_and_specialized_gates(make, fri_lde, cap):
    """Identical proofs, not only accepted ones: the oracle prover adds op-list gates (general-purpose and specialized
    placement) and the Poseidon2 flattened gate from the programs' numpy semantics; the HIP prover runs the interpreter
    kernel / the hand-written evaluator.  Every cap, opening, FRI layer and query must agree."""
    from test_gpu_prover import _compare
    from oracle import verifier as OV
    c = make()
    osetup = OP.Setup(c, fri_lde, cap, threads=8)
    po = OP.prove(c, osetup, fri_lde, cap, security_level=30, threads=8)
    gsetup = E.ProverSetup(ctx(), c, fri_lde, cap, 30)
    assert np.array_equal(gsetup.cap(), osetup.cap)
    buf, _ = gsetup.prove()
    pg = proof_format.parse(buf, security_level=30)
    _compare(pg, po)
    assert OV.verify(OV.VerificationKey(c, gsetup.cap(), fri_lde, cap), pg)
    gsetup.close()


def test_generated_kernels_and_interpreter_give_the_same_proof(tmp_path):
    """The straight-line kernels generated from the op lists (csrc/gate_aot.hip, the default) against the interpreter
    (BJ_GATE_NO_AOT=1, read once per process: a child process proves the same circuit) — identical proof bytes."""
    import os
    import subprocess
    import sys
    c = S.sha_shaped_circuit(9, seed=41, table_bits=2, extended=True)
    gsetup = E.ProverSetup(ctx(), c, 8, 16, 30)
    mine, _ = gsetup.prove()
    gsetup.close()
    out = os.path.join(str(tmp_path), "proof.npy")
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); import torch; torch.cuda.init();"
            "import era_boojum_amd as E; from era_boojum_amd import synthetic as S;"
            "c = S.sha_shaped_circuit(9, seed=41, table_bits=2, extended=True); s = E.ProverSetup(E.Context(0), c, 8, 16, 30);"
            "np.save(%r, s.prove()[0])") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                           os.path.dirname(os.path.abspath(__file__)), out)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BJ_GATE_NO_AOT="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert np.array_equal(np.load(out), mine)


def test_witness_columns_are_committed_opened_and_read_by_an_op_list_gate():
    """WitnessSet::witness (witness.rs:25): non-copiable columns behind the variables in the witness oracle
    (leaf = variables || witness || multiplicities, prover.rs:317-347), opened after the variables, outside the copy
    permutation; ZeroCheckGate with use_witness_column_for_inversion (zero_check.rs:76-161) keeps its inverse there.  The HIP
    proof equals the oracle prover's, the verifier restatement accepts it, and a wrong witness cell is refused."""
    from era_boojum_amd import synthetic as S
    c = S.sha_shaped_circuit(10, seed=5, table_bits=2, gates=S.witness_gates(60, 4, 5), mix=(0.05, 0.3, 0.3, 0.2), num_witness_cols=5)
    S.check_satisfied(c)
    assert c.num_witness_cols == 5 and any(getattr(g, "wit_stride", 0) for g in c.gates)
    osetup = OP.Setup(c, 8, 16, threads=8)
    po = OP.prove(c, osetup, 8, 16, security_level=30, threads=8)
    gsetup = E.ProverSetup(ctx(), c, 8, 16, 30)
    assert np.array_equal(gsetup.cap(), osetup.cap)
    buf, _ = gsetup.prove()
    pg = proof_format.parse(buf, security_level=30)
    assert len(pg["values_at_z"]) == len(po["values_at_z"]) and len(pg["queries_per_fri_repetition"][0]["witness_query"]["leaf_elements"]) == c.num_vars + 5 + 1
    from test_gpu_prover import _compare
    _compare(pg, po)
    assert OV.verify(OV.VerificationKey(c, gsetup.cap(), 8, 16), pg, verbose=True)
    # the inverse of a non-zero input is checked through the witness column
    zc = next(g for g in c.gates if g.name == "ZeroCheckGate[witness]")
    m = np.ones(c.n, dtype=bool)
    for i, bit in enumerate(zc.path):
        m &= c.constants[i] == (1 if bit else 0)
    r = int(np.flatnonzero(m & (c.variables[0] != 0))[0])
    bad_w = c.witness.copy()
    bad_w[0, r] = (int(bad_w[0, r]) + 1) % E.P
    with pytest.raises(E.BoojumHipError, match="not satisfied"):
        gsetup.prove(variables=np.concatenate([c.variables, bad_w], axis=0))
    gsetup.close()


def test_bounded_wrappers_are_the_inner_evaluator_with_fewer_repetitions():
    """BoundedEvaluatorWrapper / BoundedConstantsAllocatorGate / BoundedBooleanConstraintGate (bounded_wrapper.rs:60-67,
    bounded_constant_allocator.rs:78-92, bounded_boolean_allocator.rs:74-78) only cap num_repetitions_in_geometry; the formula is
    the inner evaluator's.  At this boundary the descriptor carries the repetition count, so a bounded gate is the same kind /
    op list with fewer repetitions: 2 constants per row instead of 4, 5 FMA instances instead of 15, a boolean gate over 7 of
    the general-purpose columns."""
    from era_boojum_amd.synthetic import GateDesc, GATE_CONSTANT_ALLOCATOR, GATE_FMA, GATE_PROGRAM, GATE_NOP, GATE_REDUCTION4
    gates = [
        GateDesc(GATE_CONSTANT_ALLOCATOR, "BoundedConstantsAllocatorGate", 1, 2, 1, 2, 1, 1, 1, True),
        GateDesc(GATE_FMA, "BoundedEvaluatorWrapper<FmaGateInBaseFieldWithoutConstant>", 3, 2, 4, 5, 4, 0, 1, True),
        GateDesc(GATE_REDUCTION4, "ReductionGate<4>", 2, 4, 5, 12, 5, 0, 1, True),
        GateDesc(GATE_PROGRAM, "BoundedBooleanConstraintGate", 2, 0, 1, 7, 1, 0, 1, True, program=GP.boolean_program()),
        GateDesc(GATE_NOP, "NopGate", 0, 0, 0, 1, 0, 0, 0, True),
    ]
    c = S.sha_shaped_circuit(10, seed=8, table_bits=2, gates=gates, mix=(0.1, 0.3, 0.3, 0.2))
    S.check_satisfied(c)
    osetup = OP.Setup(c, 8, 16, threads=8)
    po = OP.prove(c, osetup, 8, 16, security_level=30, threads=8)
    gsetup = E.ProverSetup(ctx(), c, 8, 16, 30)
    buf, _ = gsetup.prove()
    pg = proof_format.parse(buf, security_level=30)
    from test_gpu_prover import _compare
    _compare(pg, po)
    assert OV.verify(OV.VerificationKey(c, gsetup.cap(), 8, 16), pg, verbose=True)
    gsetup.close()


def test_dumps_with_witness_columns_through_the_c_abi():
    """bj_prove_from_dumps with a DenseWitnessCopyHint: the non-copiable witness columns are materialised from the same
    all_values by their own hint (witness.rs:445-490) and land behind the variables; same proof as the in-memory circuit."""
    import copy
    from era_boojum_amd import memcopy_format as M
    c = S.sha_shaped_circuit(10, seed=5, table_bits=2, gates=S.witness_gates(60, 4, 5), mix=(0.05, 0.3, 0.3, 0.2), num_witness_cols=5)
    a = E.ProverSetup(ctx(), c, 8, 16, 30)
    pa, _ = a.prove()
    V, n = c.variables.shape
    all_values = np.concatenate([c.variables.reshape(-1), c.witness.reshape(-1)])
    var_ids = np.arange(V * n, dtype=np.int64).reshape(V, n)
    wit_ids = V * n + np.arange(5 * n, dtype=np.int64).reshape(5, n)
    var_ids[c.variables == 0] = -1                         # empty cells: placeholders materialise as 0
    wit_dump = M.write_witness_vec([(col, row) for col, row, _ in c.public_inputs], all_values,
                                   c.multiplicities[0, :c.total_tables_len].astype(np.uint32))
    bare = copy.copy(c)
    bare.gates = [copy.copy(g) for g in c.gates]
    for g in bare.gates:
        g.path = []
    b = E.ProverSetup(ctx(), bare, 8, 16, 30, setup_base_dump=M.write_setup_base(c))
    assert np.array_equal(a.cap(), b.cap())
    pb, _ = b.prove_from_dumps(wit_dump, M.write_variables_hint(var_ids), M.write_variables_hint(wit_ids))
    assert np.array_equal(pa, pb)
    with pytest.raises(E.BoojumHipError, match="DenseWitnessCopyHint"):
        b.prove_from_dumps(wit_dump, M.write_variables_hint(var_ids))
    a.close(); b.close()


def test_specialized_gate_with_its_own_constant_columns_equals_oracle_proof():
    """Gates over specialized columns whose repetitions read their own CONSTANT columns (UseSpecializedColumns with
    share_constants = false, prover.rs:700-790: constants_for_gates_over_general_purpose_columns + initial_offset.constants_offset,
    per_repetition_offset.constants_offset): a BooleanConstraintGate over 2 columns and a ConstantsAllocatorGate over 3, whose
    constants are the last three constant columns behind the table-id column.  The HIP proof equals the oracle prover's byte
    for byte and the verifier restatement accepts it; a changed constant is reported by the prover's own check; descriptors that
    would put the constants anywhere else are refused."""
    import copy
    from oracle import prover as OP
    from oracle import verifier as OV
    from test_gpu_prover import _compare
    c = S.sha_shaped_circuit(10, seed=33, table_bits=2, boolean_columns=2, specialized_constant_columns=3)
    assert c.num_vars == 60 + 32 + 5 and c.num_constant_cols == c.table_id_col + 4 and S.check_satisfied(c)
    osetup = OP.Setup(c, 8, 16, threads=8)
    po = OP.prove(c, osetup, 8, 16, security_level=30, threads=8)
    gsetup = E.ProverSetup(ctx(), c, 8, 16, 30)
    assert np.array_equal(gsetup.cap(), osetup.cap)
    buf, _ = gsetup.prove()
    pg = proof_format.parse(buf, security_level=30)
    _compare(pg, po)
    assert OV.verify(OV.VerificationKey(c, gsetup.cap(), 8, 16), pg, verbose=True)
    # the golden-pinned quotient-identity code (oracle/golden_quotient.py), fed with the VerificationKey JSON this repository emits
    import json
    import oracle as O
    from oracle import golden_quotient as GQ
    from era_boojum_amd import wire_format as W
    vk = json.loads(W.dumps(W.vk_to_reference_json(c, gsetup.cap(), 8, 16)))
    t = O.Transcript()
    t.absorb_cap(gsetup.cap())
    t.absorb(pg["public_inputs"])
    t.absorb_cap(np.array(pg["witness_oracle_cap"], dtype=np.uint64))
    beta, gamma, lbeta, lgamma = (t.challenge_ext() for _ in range(4))
    t.absorb_cap(np.array(pg["stage_2_oracle_cap"], dtype=np.uint64))
    alpha = t.challenge_ext()
    t.absorb_cap(np.array(pg["quotient_oracle_cap"], dtype=np.uint64))
    z = t.challenge_ext()
    lhs, rhs = GQ.quotient_identity(GQ.geometry_from_vk_json(vk), [g.name for g in c.gates],
                                    [("BooleanConstraintGate", 2), ("ConstantsAllocatorGate", 3)], c.non_residues,
                                    dict(beta=beta, gamma=gamma, lookup_beta=lbeta, lookup_gamma=lgamma, alpha=alpha, z=z),
                                    pg["values_at_z"], pg["values_at_z_omega"][0])
    assert lhs == rhs
    gsetup.close()
    bad = copy.copy(c)                                  # the same cells against another constant: the quotient is not a polynomial
    bad.constants = c.constants.copy()
    bad.constants[-1, 5] ^= np.uint64(1)
    bsetup = E.ProverSetup(ctx(), bad, 8, 16, 30)
    with pytest.raises(E.BoojumHipError, match="not satisfied"):
        bsetup.prove()
    bsetup.close()
    wrong = copy.copy(c)                                # constant columns that do not add up to num_constant_cols
    wrong.specialized_gates = [copy.copy(g) for g in c.specialized_gates]
    wrong.specialized_gates[1].const_stride = 2
    with pytest.raises(E.BoojumHipError, match="constant columns declared"):
        E.ProverSetup(ctx(), wrong, 8, 16, 30)
    shared = copy.copy(c)                               # share_constants = true with constants: refused, as the reference's own
    shared.specialized_gates = [copy.copy(g) for g in c.specialized_gates]   # prover hands such an evaluator an empty range
    shared.specialized_gates[1].const_stride = 0
    shared.constants = c.constants[:-3]
    shared.num_constant_cols = c.num_constant_cols - 3
    with pytest.raises(E.BoojumHipError, match="share_constants = false"):
        E.ProverSetup(ctx(), shared, 8, 16, 30)
