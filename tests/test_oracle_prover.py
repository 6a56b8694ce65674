"""End-to-end oracle tests: a satisfiable SHA-shaped synthetic circuit is proven by the CPU restatement of
prove_cpu_basic (oracle/prover.py) and the proof is accepted by the restatement of Verifier::verify
(oracle/verifier.py); corrupted witnesses / proofs are rejected (the reference's own test pattern 2:
"verifier accepts", cs.rs:1076, 1465)."""
import copy

import numpy as np
import pytest

import oracle as O
from era_boojum_amd import synthetic as S
from oracle import prover as OP
from oracle import verifier as OV


@pytest.fixture(scope="module")
def small_case():
    c = S.sha_shaped_circuit(9, seed=7, table_bits=2)
    assert S.check_satisfied(c)
    fri_lde, cap = 8, 16
    setup = OP.Setup(c, fri_lde, cap, threads=4)
    proof, aux = OP.prove(c, setup, fri_lde, cap, security_level=30, threads=4, return_aux=True)
    vk = OV.VerificationKey(c, setup.cap, fri_lde, cap)
    return c, setup, proof, aux, vk


def test_verifier_accepts_oracle_proof(small_case):
    c, setup, proof, aux, vk = small_case
    assert OV.verify(vk, proof, verbose=True)
    assert len(proof["values_at_z"]) == 92 + 8 + 92 + 1 + 22 + (1 + 8 + 1 + 5) + 4     # SURVEY §8 counts
    assert len(proof["queries_per_fri_repetition"][0]["witness_query"]["leaf_elements"]) == 93
    assert len(proof["queries_per_fri_repetition"][0]["stage_2_query"]["leaf_elements"]) == 64
    assert len(proof["queries_per_fri_repetition"][0]["setup_query"]["leaf_elements"]) == 105


def test_grand_product_closes_and_quotient_is_low_degree(small_case):
    c, setup, proof, aux, vk = small_case
    # z(omega^n) = z(1) = 1: the shifted grand product returns to one (copy_permutation.rs:484)
    z = aux["z_nat"]
    assert int(z[0][0]) == 1 and int(z[1][0]) == 0
    # quotient monomials: degree < q*n - 1 and not trivially zero
    qm = aux["qmono"]
    assert qm[0][-1] == 0 and qm[1][-1] == 0 and qm.any()


def test_verifier_rejects_tampering(small_case):
    c, setup, proof, aux, vk = small_case
    bad = copy.deepcopy(proof)
    bad["values_at_z"][3][0] = (bad["values_at_z"][3][0] + 1) % O.P
    assert not OV.verify(vk, bad)
    bad = copy.deepcopy(proof)
    bad["queries_per_fri_repetition"][0]["witness_query"]["leaf_elements"][5] ^= 1
    assert not OV.verify(vk, bad)
    bad = copy.deepcopy(proof)
    bad["final_fri_monomials"][0][0] = (bad["final_fri_monomials"][0][0] + 1) % O.P
    assert not OV.verify(vk, bad)
    bad = copy.deepcopy(proof)
    bad["public_inputs"][0] = (bad["public_inputs"][0] + 1) % O.P
    assert not OV.verify(vk, bad)


def test_unsatisfied_witness_is_caught():
    c = S.sha_shaped_circuit(8, seed=3, table_bits=2)
    fma = [g for g in c.gates if g.kind == S.GATE_FMA][0]
    rows = np.nonzero(c.constants[0] == 1)[0]          # FMA rows have path [1]
    c.variables[3, rows[0]] = (int(c.variables[3, rows[0]]) + 1) % O.P   # break one FMA output
    setup = OP.Setup(c, 8, 16, threads=4)
    with pytest.raises(AssertionError):
        OP.prove(c, setup, 8, 16, security_level=20, threads=4)


def test_fri_lde_2_config_like_the_golden_proof():
    """fri_lde_factor 2 < quotient degree 4 (the golden proof's regime: LDE for the quotient is wider than the FRI rate)."""
    c = S.sha_shaped_circuit(8, seed=11, table_bits=2)
    setup = OP.Setup(c, 2, 4, threads=4)
    proof = OP.prove(c, setup, 2, 4, security_level=20, threads=4)
    assert OV.verify(OV.VerificationKey(c, setup.cap, 2, 4), proof, verbose=True)


def test_coset_streaming_restatement_equals_the_full_prover(small_case):
    """oracle/prover_streaming.py (every LDE one coset at a time: what lets the 2^22-row bench circuit be checked within a
    test's memory) gives the caps and openings of oracle/prover.py, under both algebraic transcripts."""
    from oracle import prover_streaming as PS
    c, setup, proof, aux, vk = small_case
    for kind in (1, 2):
        want = proof if kind == 1 else OP.prove(c, setup, 8, 16, security_level=30, threads=4, transcript_kind=2)
        got = PS.commitments_and_openings(c, setup.cap, 8, 16, threads=4, transcript_kind=kind, check_setup_cosets=(0, 3, 7))
        for k in ("public_inputs", "witness_oracle_cap", "stage_2_oracle_cap", "quotient_oracle_cap", "values_at_z", "values_at_z_omega",
                  "values_at_0"):
            assert got[k] == want[k], (kind, k)
        for cs, frag in got["setup_cap_fragments"].items():
            assert np.array_equal(frag, setup.cap[2 * cs:2 * cs + 2])
    assert got["challenges"]["z"] != aux["z"]      # the other transcript draws other challenges: the comparison above is not vacuous


def test_coset_streaming_restatement_of_the_rest_of_the_proof(small_case):
    """Past the openings (what pins the WHOLE transcript at the bench's 2^22 rows on hosts that cannot hold the full oracle
    prover): the DEEP accumulator coset by coset, do_fri on it, the queries.  Against oracle/prover.py under both algebraic
    transcripts: DEEP challenge, every FRI cap and fold challenge, final monomials, query indices, every FRI query opening byte
    for byte; for the witness / second-stage / quotient oracles every query's path and the hash of the full prover's opened
    leaf; the setup oracle's on the cosets asked for.  A proof with one changed opened element or one changed FRI leaf is
    told apart."""
    from oracle import prover_streaming as PS
    c, setup, proof, aux, vk = small_case
    N, n = c.n * 8, c.n
    for kind in (1, 2):
        want = proof if kind == 1 else OP.prove(c, setup, 8, 16, security_level=30, threads=4, transcript_kind=2)
        got = PS.commitments_and_openings(c, setup.cap, 8, 16, threads=4, transcript_kind=kind, check_setup_cosets=(0, 3, 7),
                                          rest_of_the_proof=True, security_level=30)
        compared = PS.compare_rest_of_the_proof(want, got)
        nq = len(want["queries_per_fri_repetition"])
        in_setup = sum(1 for idx in got["query_indexes"] if (idx >> c.log_n) in (0, 3, 7))
        assert compared == 3 * nq + in_setup and 0 < in_setup < nq
        if kind == 1:
            assert got["deep_challenge"] == aux["deep_challenge"] and got["fri_challenges"] == aux["fri_challenges"]
            fri_slot = aux["deep"][0][got["query_indexes"][0]]
            assert want["queries_per_fri_repetition"][0]["fri_queries"][0]["leaf_elements"][got["query_indexes"][0] & 7] == int(fri_slot)
    bad = copy.deepcopy(proof)
    bad["queries_per_fri_repetition"][2]["stage_2_query"]["leaf_elements"][11] ^= 1
    got = PS.commitments_and_openings(c, setup.cap, 8, 16, threads=4, rest_of_the_proof=True, security_level=30)
    with pytest.raises(AssertionError, match="stage_2_query leaf"):
        PS.compare_rest_of_the_proof(bad, got)
    bad = copy.deepcopy(proof)
    bad["queries_per_fri_repetition"][1]["fri_queries"][1]["leaf_elements"][3] ^= 1
    with pytest.raises(AssertionError, match="FRI query"):
        PS.compare_rest_of_the_proof(bad, got)
    bad = copy.deepcopy(proof)
    bad["queries_per_fri_repetition"][0]["witness_query"]["proof"][4][0] ^= 1
    with pytest.raises(AssertionError, match="witness_query path"):
        PS.compare_rest_of_the_proof(bad, got)
    # claimed-cap mode (2^23 rows): only two cosets of the two big oracles are hashed; the rest of the proof is still recomputed
    claimed = {k: proof[k] for k in ("witness_oracle_cap", "stage_2_oracle_cap", "quotient_oracle_cap")}
    got = PS.commitments_and_openings(c, setup.cap, 8, 16, threads=4, cap_cosets=(0, 5), claimed_caps=claimed, check_setup_cosets=(0, 5),
                                      rest_of_the_proof=True, security_level=30)
    compared = PS.compare_rest_of_the_proof(proof, got)
    in_two = sum(1 for idx in got["query_indexes"] if (idx >> c.log_n) in (0, 5))
    assert compared == len(got["query_indexes"]) + 3 * in_two      # quotient everywhere; witness, stage 2, setup on cosets {0, 5}


def test_coset_streaming_restatement_with_claimed_caps(small_case):
    """The cfg5 mode (2^23 rows): only cosets {0, 5} of the witness and second-stage oracles are hashed, the transcript absorbs
    the caps of the proof under check, the recomputed subtree roots are returned for comparison and every opening is still
    recomputed.  A proof with one wrong cap node is told apart: its fragment differs, and so does everything drawn after it."""
    from oracle import prover_streaming as PS
    c, setup, proof, aux, vk = small_case
    claimed = {k: proof[k] for k in ("witness_oracle_cap", "stage_2_oracle_cap", "quotient_oracle_cap")}
    got = PS.commitments_and_openings(c, setup.cap, 8, 16, threads=4, cap_cosets=(0, 5), claimed_caps=claimed)
    for k in ("quotient_oracle_cap", "values_at_z", "values_at_z_omega", "values_at_0"):
        assert got[k] == proof[k], k
    for name in ("witness_oracle_cap", "stage_2_oracle_cap"):
        assert sorted(got["cap_fragments"][name]) == [0, 5]
        for cs, frag in got["cap_fragments"][name].items():
            assert np.array_equal(frag, np.asarray(proof[name], dtype=np.uint64)[2 * cs:2 * cs + 2]), (name, cs)
    bad = dict(claimed)
    bad["witness_oracle_cap"] = [list(x) for x in claimed["witness_oracle_cap"]]
    bad["witness_oracle_cap"][10][1] ^= 1
    got = PS.commitments_and_openings(c, setup.cap, 8, 16, threads=4, cap_cosets=(5,), claimed_caps=bad)
    assert not np.array_equal(got["cap_fragments"]["witness_oracle_cap"][5], np.asarray(bad["witness_oracle_cap"], dtype=np.uint64)[10:12])
    assert got["values_at_z"] != proof["values_at_z"]


def test_specialized_gate_with_its_own_constant_columns():
    """GatePlacementStrategy::UseSpecializedColumns { share_constants: false } for an evaluator that reads a constant inside
    evaluate_once (ConstantsAllocatorGate): every repetition owns a variable column and a CONSTANT column; the constant columns
    are the last ones, behind the general-purpose gates' constants and the table-id column (evaluator_data.rs:196-238,
    prover.rs:748-772, verifier.rs:1609-1633).  The oracle prover's proof is accepted by the verifier restatement; a
    verifier that places those constants one column off rejects it, and a changed constant makes the witness unsatisfying."""
    from era_boojum_amd import synthetic as S
    c = S.sha_shaped_circuit(8, seed=91, table_bits=2, boolean_columns=2, specialized_constant_columns=3)
    assert c.num_vars == 60 + 32 + 2 + 3 and c.num_constant_cols == c.table_id_col + 1 + 3
    assert [g.name for g in c.specialized_gates] == ["BooleanConstraintGate", "ConstantsAllocatorGate"]
    assert S.check_satisfied(c)
    setup = OP.Setup(c, 8, 16, threads=4)
    proof = OP.prove(c, setup, 8, 16, security_level=20, threads=4)
    vk = OV.VerificationKey(c, setup.cap, 8, 16)
    assert OV.verify(vk, proof, verbose=True)
    # the quotient-identity code that the reference's own proof pins (oracle/golden_quotient.py), in the reference's VK layout
    import json
    import oracle as O
    from era_boojum_amd import wire_format as W
    from oracle import golden_quotient as GQ
    vkj = json.loads(W.dumps(W.vk_to_reference_json(c, setup.cap, 8, 16)))
    t = O.Transcript()
    t.absorb_cap(setup.cap)
    t.absorb(proof["public_inputs"])
    t.absorb_cap(np.array(proof["witness_oracle_cap"], dtype=np.uint64))
    beta, gamma, lbeta, lgamma = (t.challenge_ext() for _ in range(4))
    t.absorb_cap(np.array(proof["stage_2_oracle_cap"], dtype=np.uint64))
    alpha = t.challenge_ext()
    t.absorb_cap(np.array(proof["quotient_oracle_cap"], dtype=np.uint64))
    z = t.challenge_ext()
    lhs, rhs = GQ.quotient_identity(GQ.geometry_from_vk_json(vkj), [g.name for g in c.gates],
                                    [("BooleanConstraintGate", 2), ("ConstantsAllocatorGate", 3)], c.non_residues,
                                    dict(beta=beta, gamma=gamma, lookup_beta=lbeta, lookup_gamma=lgamma, alpha=alpha, z=z),
                                    proof["values_at_z"], proof["values_at_z_omega"][0])
    assert lhs == rhs
    import copy
    swapped = dict(proof)                                       # the openings of the last two constant columns exchanged: the
    vz = [list(v) for v in proof["values_at_z"]]                # repetitions then read each other's constant
    i = c.num_vars + c.num_constant_cols - 1
    vz[i], vz[i - 1] = vz[i - 1], vz[i]
    swapped["values_at_z"] = vz
    assert not OV.verify(vk, swapped)
    bad = copy.copy(c)
    bad.constants = c.constants.copy()
    bad.constants[-2, 17] ^= np.uint64(1)
    with pytest.raises(AssertionError):
        S.check_satisfied(bad)
    with pytest.raises(AssertionError, match="unsatisfied"):
        OP.prove(bad, OP.Setup(bad, 8, 16, threads=4), 8, 16, security_level=20, threads=4)


def test_lookup_with_the_table_id_as_a_variable_column():
    """UseSpecializedColumnsWithTableIdAsVariable (cs/mod.rs:237-241): the CPU restatement of the second mode of
    compute_lookup_poly_pairs_specialized / compute_quotient_terms_for_lookup_specialized (lookup_argument_in_ext.rs:354-366,
    949-1000) proves a circuit whose sub-arguments each carry their own table id in a (width+1)-th variable column; the verifier
    restatement (verifier.rs:1402-1464) and the quotient-identity code pinned by the reference's own proof accept it, the
    coset-streaming restatement gives the same caps and openings, and a wrong table id breaks the verifier's lookup sumcheck."""
    import json
    from era_boojum_amd import wire_format as W
    from oracle import golden_quotient as GQ
    from oracle import prover_streaming as PS
    c = S.sha_shaped_circuit(9, seed=17, table_bits=2, table_id_as_variable=True, boolean_columns=2)
    assert S.check_satisfied(c) and c.num_vars == 60 + 8 * 5 + 2 and c.num_constant_cols == c.num_constants_for_gates
    ids = c.variables[c.num_gp_vars + 4::5][:8]
    assert not np.array_equal(ids[0], ids[1]) and 1 <= int(ids.min()) and int(ids.max()) <= 5      # a table per sub-argument and row
    setup = OP.Setup(c, 8, 16, threads=4)
    proof = OP.prove(c, setup, 8, 16, security_level=30, threads=4)
    vk = OV.VerificationKey(c, setup.cap, 8, 16)
    assert OV.verify(vk, proof, verbose=True)
    assert len(proof["values_at_z"]) == 102 + c.num_constant_cols + 102 + 1 + 25 + (1 + 8 + 1 + 5) + 4
    assert len(proof["queries_per_fri_repetition"][0]["witness_query"]["leaf_elements"]) == 103
    # the identity code that holds on the reference's own proof, from the VerificationKey JSON of this circuit
    vkj = json.loads(W.dumps(W.vk_to_reference_json(c, setup.cap, 8, 16)))
    assert vkj["fixed_parameters"]["lookup_parameters"] == {"UseSpecializedColumnsWithTableIdAsVariable":
                                                            {"width": 4, "num_repetitions": 8, "share_table_id": False}}
    t = O.Transcript()
    t.absorb_cap(setup.cap)
    t.absorb(proof["public_inputs"])
    t.absorb_cap(np.array(proof["witness_oracle_cap"], dtype=np.uint64))
    beta, gamma, lbeta, lgamma = (t.challenge_ext() for _ in range(4))
    t.absorb_cap(np.array(proof["stage_2_oracle_cap"], dtype=np.uint64))
    alpha = t.challenge_ext()
    t.absorb_cap(np.array(proof["quotient_oracle_cap"], dtype=np.uint64))
    z = t.challenge_ext()
    lhs, rhs = GQ.quotient_identity(GQ.geometry_from_vk_json(vkj), [g.name for g in c.gates], [("BooleanConstraintGate", 2)],
                                    c.non_residues, dict(beta=beta, gamma=gamma, lookup_beta=lbeta, lookup_gamma=lgamma, alpha=alpha, z=z),
                                    proof["values_at_z"], proof["values_at_z_omega"][0])
    assert lhs == rhs
    bad = copy.deepcopy(proof)                              # the opening of a table-id column at z
    k = c.num_gp_vars + 4
    bad["values_at_z"][k] = [(bad["values_at_z"][k][0] + 1) % O.P, bad["values_at_z"][k][1]]
    assert not OV.verify(vk, bad)
    plain = S.sha_shaped_circuit(8, seed=18, table_bits=2, table_id_as_variable=True)      # the streaming restatement's circuit class
    psetup = OP.Setup(plain, 8, 16, threads=4)
    want = OP.prove(plain, psetup, 8, 16, security_level=20, threads=4)
    got = PS.commitments_and_openings(plain, psetup.cap, 8, 16, threads=4)
    for key in ("witness_oracle_cap", "stage_2_oracle_cap", "quotient_oracle_cap", "values_at_z", "values_at_z_omega", "values_at_0"):
        assert got[key] == want[key], key
    wrong = S.sha_shaped_circuit(8, seed=3, table_bits=2, table_id_as_variable=True)
    col = wrong.num_gp_vars + 2 * 5 + 4
    wrong.variables[col, 9] = np.uint64(77)                 # names a table that does not exist: the tuple is in no table
    # A_i = 1 / denominator makes every lookup TERM of the quotient vanish by construction; what a wrong tuple breaks is the sum
    # sum_i A_i(0) = B(0) the verifier checks (verifier.rs:1236-1256), exactly as in the reference (its prover sums only under
    # DEBUG_SATISFIABLE, lookup_argument_in_ext.rs:672-700)
    wsetup = OP.Setup(wrong, 8, 16, threads=4)
    wproof = OP.prove(wrong, wsetup, 8, 16, security_level=20, threads=4)
    assert not OV.verify(OV.VerificationKey(wrong, wsetup.cap, 8, 16), wproof)
